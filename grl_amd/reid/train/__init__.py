from .trainer import SEQTrainer
