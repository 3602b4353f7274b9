"""Oracle: triggers, targets, masks, poisoned-sample blend (numpy / torch CPU).

Follows /root/reference/dataset.py:504-524 (box trigger), :526-597 (get_trigger),
:627-655 (get_target), :447-450 (__bg2grey), :499-503 (__roll), :275-276
(get_mask), :288-315 (clean / backdoor transforms) and util.py:83-111
(normalize, eps = 1e-5).  Box triggers, CORNER / TRIGGER / SHIFT targets and the
int mask are pinned by tests/golden/backdoor.npz.  Image-file triggers/targets
(GLASSES, STOP_SIGN_*, HAT, CAT) need torchvision, which the container lacks:
their PIL restatement below is "parity unpinned".
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import numpy as np
import torch

VMIN, VMAX = -1.0, 1.0
GREY_BG_RATIO = 0.3
GAP = 2                      # dataset.py:409 TRIGGER_GAP_X = TRIGGER_GAP_Y = 2

BOX_SIZES = {"SM_BOX": (14, "white"), "XSM_BOX": (11, "white"), "XXSM_BOX": (8, "white"),
             "XXXSM_BOX": (4, "white"), "BIG_BOX": (18, "white"), "BOX_18": (18, "grey"),
             "BOX_14": (14, "grey"), "BOX_11": (11, "grey"), "BOX_8": (8, "grey"), "BOX_4": (4, "grey")}


def normalize(x, vmin_in=0.0, vmax_in=1.0, vmin_out=VMIN, vmax_out=VMAX, eps=1e-5):
    # util.py:111
    return ((x - vmin_in) / (vmax_in - vmin_in + eps)) * (vmax_out - vmin_out) + vmin_out


def image_u8_to_float(u8_hwc):
    """uint8 [H,W,C] -> float32 [C,H,W] in [-1, 0.99998]: ToTensor (/255) then normalize (dataset.py:120-136)."""
    x = torch.from_numpy(np.asarray(u8_hwc)).permute(2, 0, 1).to(torch.float32) / 255.0
    return normalize(x)


def box_trigger(k, kind, channel, image_size, vmin=VMIN, vmax=VMAX):
    # dataset.py:504-524: rows/cols [-(k+2):-2] set to val on a vmin background
    val = vmax if kind == "white" else (vmin + vmax) / 2
    trig = torch.full((channel, image_size, image_size), float(vmin))
    trig[:, -(k + GAP):-GAP, -(k + GAP):-GAP] = val
    return trig


def bg2grey(t, vmin=VMIN, vmax=VMAX):
    thres = (vmax - vmin) * GREY_BG_RATIO + vmin          # dataset.py:447-450 -> -0.4
    t = t.clone()
    t[t <= thres] = thres
    return t


def get_trigger(kind, channel, image_size):
    if kind in BOX_SIZES:
        k, colour = BOX_SIZES[kind]
        return box_trigger(k, colour, channel, image_size)
    if kind == "NONE":
        return torch.full((channel, image_size, image_size), VMIN)
    raise ValueError(f"Trigger type {kind} isn't found")


def get_target(kind, trigger, dx=-5, dy=-3):
    channel, image_size = trigger.shape[0], trigger.shape[-1]
    if kind == "TRIGGER":
        return bg2grey(trigger)
    if kind == "SHIFT":
        return bg2grey(torch.roll(trigger, shifts=(0, dy, dx), dims=(0, 1, 2)))      # dataset.py:499-503
    if kind == "CORNER":
        t = torch.full((channel, image_size, image_size), VMIN)
        t[:, :10, :10] = (VMIN + VMAX) / 2                                             # dataset.py:641-644
        return bg2grey(t)
    raise NotImplementedError(f"Target type {kind} isn't found")


def get_mask(trigger, vmin=VMIN):
    # dataset.py:275-276: int64, per element (channel included)
    return torch.where(trigger > vmin, 0, 1)


def make_batch(images, is_poison, trigger, target):
    """images float [B,C,H,W] (already normalised); is_poison bool [B].
    Returns (R = pixel_values, x0 = target) exactly as the collated DataLoader batch
    (dataset.py:288-315): clean rows R = 0, x0 = x; poisoned rows R = m*x + (1-m)*g, x0 = y."""
    m = get_mask(trigger).to(images.dtype)
    poisoned = m * images + (1 - m) * trigger
    sel = is_poison.reshape(-1, 1, 1, 1)
    R = torch.where(sel, poisoned, torch.zeros_like(images))
    x0 = torch.where(sel, target.expand_as(images), images)
    return R, x0
