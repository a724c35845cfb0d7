from .eva_functions import accuracy, cmc, mean_ap
from .attevaluator import ATTEvaluator
from .rerank import re_ranking

__all__ = ['accuracy', 'cmc', 'mean_ap', 're_ranking', 'ATTEvaluator']
