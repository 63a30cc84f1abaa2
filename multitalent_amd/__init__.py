"""multitalent_amd — MI355X-native hot path for MultiTalent / nnU-Net v1 style 3D patch segmentation.

Python here is orchestration only; all arithmetic runs in libmtseg_hip.so (include/mtseg.h).
"""
__version__ = '0.1.0'
