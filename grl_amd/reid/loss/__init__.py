from .oim import oim, OIM, OIMLoss
from .pairloss import PairLoss
from .triplet import TripletLoss, TripletLoss_OIM

__all__ = ['oim', 'OIM', 'OIMLoss', 'PairLoss', 'TripletLoss', 'TripletLoss_OIM']
