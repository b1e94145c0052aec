"""segmentation_amd: MI355X-native U-Net / FCN train-step + inference hot path behind the
UNetModel / FCNModel / BaseModel surface of nathanin/segmentation."""
from ._lib import SegError  # noqa: F401


def __getattr__(name):
    if name == 'UNetModel':
        from .unet import UNetModel
        return UNetModel
    if name == 'FCNModel':
        from .fcn import FCNModel
        return FCNModel
    if name == 'DeconvModel':
        from .deconvolution import DeconvModel
        return DeconvModel
    if name == 'BaseModel':
        from .basemodel import BaseModel
        return BaseModel
    raise AttributeError(name)
