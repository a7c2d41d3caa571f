"""Model registry (reference: models/__init__.py:4-32; synonyms options/base_options.py:223-233)."""
from .base_model import BaseModel

_SYNONYMS = {"gmm": "warp", "tom": "unet_mask", "unet": "unet_mask"}


def canonical_model_name(name):
    name = name.lower()
    return _SYNONYMS.get(name, name)


def find_model_using_name(model_name):
    model_name = canonical_model_name(model_name)
    if model_name == "warp":
        from . import warp_model as lib
    elif model_name == "unet_mask":
        from . import unet_mask_model as lib
    elif model_name == "sams":
        from . import sams_model as lib
    else:
        raise NotImplementedError(
            f"model '{model_name}' is outside the MI355X hot path (warp / unet_mask / sams); see DESIGN.md"
        )
    target = model_name.replace("_", "") + "model"
    for name, cls in lib.__dict__.items():
        if name.lower() == target and isinstance(cls, type) and issubclass(cls, BaseModel):
            return cls
    raise NotImplementedError(f"no BaseModel subclass named {target} in {lib.__name__}")


def get_option_setter(model_name):
    return find_model_using_name(model_name).modify_commandline_options
