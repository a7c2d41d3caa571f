"""Channel bookkeeping of the batch dict (reference: datasets/tryon_dataset.py:47-61,540-547)."""
RGB_CHANNELS = 3
MASK_CHANNELS = 1
COCOPOSE_CHANNELS = 18
IM_HEAD_CHANNELS = RGB_CHANNELS
SILHOUETTE_CHANNELS = MASK_CHANNELS
AGNOSTIC_CHANNELS = IM_HEAD_CHANNELS + SILHOUETTE_CHANNELS
CLOTH_CHANNELS = RGB_CHANNELS
CLOTH_MASK_CHANNELS = MASK_CHANNELS
DENSEPOSE_CHANNELS = 3
FLOW_CHANNELS = 2
IMAGE_CHANNELS = RGB_CHANNELS
IM_CLOTH_CHANNELS = RGB_CHANNELS
GRID_VIS_CHANNELS = RGB_CHANNELS


def parse_num_channels(list_of_inputs):
    """Number of channels of the concatenation of the named batch tensors."""
    if isinstance(list_of_inputs, str):
        list_of_inputs = [list_of_inputs]
    return sum(globals()[f"{inp.upper()}_CHANNELS"] for inp in list_of_inputs)
