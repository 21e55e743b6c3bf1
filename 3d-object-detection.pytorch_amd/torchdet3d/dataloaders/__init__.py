from .gpu_crops import FrameCropper, crop_cords_from_keypoints
