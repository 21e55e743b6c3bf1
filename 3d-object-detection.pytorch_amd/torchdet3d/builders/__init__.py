from .model_builder import build_model, ModelWrapper, AVAILABLE_MODELS
from .loss_builder import build_loss, AVAILABLE_LOSS
from .optim_builder import build_optimizer, AVAILABLE_OPTIMS
from .scheduler_builder import build_scheduler, AVAILABLE_SCHEDS
from .loader_builder import build_loader, SyntheticCrops
