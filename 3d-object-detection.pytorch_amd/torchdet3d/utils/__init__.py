from .utils import (AverageMeter, AttrDict, read_py_config, save_snap, load_checkpoint, load_pretrained_weights,
                    resume_from, set_random_seed, put_on_device, check_isfile, mkdir_if_missing, Logger,
                    OBJECTRON_CLASSES)
from .geometry import lift_2d, project_3d_points, convert_2d_to_ndc, convert_camera_matrix_2_ndc, \
    get_default_camera_matrix
from .ie_wrappers import Regressor, Detector
