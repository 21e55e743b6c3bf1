from .train import Trainer
