"""Backdoor triggers / targets and the poisoned DatasetLoader -- drop-in for /root/reference/dataset.py.

`Backdoor.get_trigger / get_target` (dataset.py:526-597, :627-655) are host-side constructors (tiny [C,H,W]
tensors, built once); the per-sample work of the reference's DataLoader workers -- ToTensor + normalize
(dataset.py:120-136, util.py:83-111), random H-flip, get_mask (:275-276), blend (:306-315) -- happens ON THE
DEVICE: the whole uint8 dataset lives in HBM (CIFAR10: 184 MB) and `poison_qsample` consumes uint8 batches
directly (SURVEY f-1).  `DatasetLoader.get_dataloader()` still yields the reference's dict batches
(`pixel_values`, `target`, `image`, `label`, `is_clean`) for code that iterates them.

Offline policy: HF-hub datasets are unreachable here; a dataset is either a local .npy/.npz of uint8 images
([N,H,W,C]) under `root`, or synthetic (seeded uniform uint8), stated in `DatasetLoader.source`.
"""
import os
from typing import Optional

import numpy as np
import torch

DEFAULT_VMIN = float(-1.0)
DEFAULT_VMAX = float(1.0)


def normalize(x, vmin_in=None, vmax_in=None, vmin_out=0, vmax_out=1, eps=1e-5):
    """util.py:83-111"""
    if vmax_out is None and vmin_out is None:
        return x
    min_x = x.min() if vmin_in is None else vmin_in
    max_x = x.max() if vmax_in is None else vmax_in
    if vmax_out is None:
        vmax_out = max_x
    if vmin_out is None:
        vmin_out = min_x
    return ((x - min_x) / (max_x - min_x + eps)) * (vmax_out - vmin_out) + vmin_out


class Backdoor:
    CHANNEL_LAST = -1
    CHANNEL_FIRST = -3
    GREY_BG_RATIO = 0.3
    STOP_SIGN_IMG = "static/stop_sign_wo_bg.png"
    CAT_IMG = "static/cat_wo_bg.png"
    GLASSES_IMG = "static/glasses.png"
    HAT_IMG = "static/fedora-hat.png"

    TARGET_SHOE = "SHOE"
    TARGET_TG = "TRIGGER"
    TARGET_CORNER = "CORNER"
    TARGET_SHIFT = "SHIFT"
    TARGET_HAT = "HAT"
    TARGET_CAT = "CAT"

    TRIGGER_GAP_X = TRIGGER_GAP_Y = 2

    TRIGGER_NONE = "NONE"
    TRIGGER_SM_BOX = "SM_BOX"
    TRIGGER_XSM_BOX = "XSM_BOX"
    TRIGGER_XXSM_BOX = "XXSM_BOX"
    TRIGGER_XXXSM_BOX = "XXXSM_BOX"
    TRIGGER_BIG_BOX = "BIG_BOX"
    TRIGGER_BOX_18 = "BOX_18"
    TRIGGER_BOX_14 = "BOX_14"
    TRIGGER_BOX_11 = "BOX_11"
    TRIGGER_BOX_8 = "BOX_8"
    TRIGGER_BOX_4 = "BOX_4"
    TRIGGER_GLASSES = "GLASSES"
    TRIGGER_STOP_SIGN_18 = "STOP_SIGN_18"
    TRIGGER_STOP_SIGN_14 = "STOP_SIGN_14"
    TRIGGER_STOP_SIGN_11 = "STOP_SIGN_11"
    TRIGGER_STOP_SIGN_8 = "STOP_SIGN_8"
    TRIGGER_STOP_SIGN_4 = "STOP_SIGN_4"

    _WHITE = {"SM_BOX": 14, "XSM_BOX": 11, "XXSM_BOX": 8, "XXXSM_BOX": 4, "BIG_BOX": 18}
    _GREY = {"BOX_18": 18, "BOX_14": 14, "BOX_11": 11, "BOX_8": 8, "BOX_4": 4}
    _STOP = {"STOP_SIGN_18": 18, "STOP_SIGN_14": 14, "STOP_SIGN_11": 11, "STOP_SIGN_8": 8, "STOP_SIGN_4": 4}

    def __init__(self, root: Optional[str]):
        self._root = root

    # ---- image-file triggers / targets: the reference's transform chain (dataset.py:427-441: convert -> transforms.Resize(int) ->
    # ToTensor -> normalize -> Pad) applied to PIL images.  On a PIL image torchvision's Resize IS PIL's: transforms.functional.resize ->
    # _compute_resized_output_size (shorter side -> size, longer = int(size * long / short)) -> img.resize((w, h), Image.BILINEAR), and ToTensor
    # is uint8 -> float32 / 255 in CHW; the same calls are made here.  torchvision itself is absent from this image, so the equality is by
    # construction, not by a run against it (unpinned) -----
    def _asset(self, rel):
        for base in (self._root, os.getcwd(), os.path.dirname(os.path.dirname(os.path.abspath(__file__)))):
            if base and os.path.exists(os.path.join(base, rel)):
                return os.path.join(base, rel)
        raise FileNotFoundError(
            f"{rel} not found: image triggers/targets need the reference's static/ assets next to the run "
            "(they are not redistributed with this package)")

    @staticmethod
    def _load_img(path, channel, size, vmin, vmax):
        from PIL import Image
        img = Image.open(path)
        img = img.convert("L") if channel == 1 else img.convert("RGB")
        if isinstance(size, int):      # transforms.Resize(int): shorter side -> size, bilinear
            w, h = img.size
            if w <= h:
                nw, nh = size, int(size * h / w)
            else:
                nh, nw = size, int(size * w / h)
        else:
            nh, nw = size
        img = img.resize((nw, nh), Image.BILINEAR)
        x = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).float() / 255.0
        x = x[None] if x.dim() == 2 else x.permute(2, 0, 1)
        return normalize(x, vmin_in=0.0, vmax_in=1.0, vmin_out=vmin, vmax_out=vmax)

    def _img_trigger(self, rel, image_size, channel, trigger_sz, vmin, vmax, x=None, y=None):
        # dataset.py:472-497
        l_pad = t_pad = int((image_size - trigger_sz) / 2)
        r_pad = image_size - trigger_sz - l_pad
        b_pad = image_size - trigger_sz - t_pad
        residual = image_size - trigger_sz
        if x is not None:
            if x > 0:
                l_pad, r_pad = x, residual - x
            else:
                r_pad = -x; l_pad = residual - r_pad
        if y is not None:
            if y > 0:
                t_pad, b_pad = y, residual - y
            else:
                b_pad = -y; t_pad = residual - b_pad
        trig = self._load_img(self._asset(rel), channel, trigger_sz, vmin, vmax)
        trig = torch.nn.functional.pad(trig, (l_pad, r_pad, t_pad, b_pad), value=vmin)
        trig[trig >= 0.999] = vmin
        return trig

    @staticmethod
    def _bg2grey(trig, vmin, vmax):
        thres = (vmax - vmin) * Backdoor.GREY_BG_RATIO + vmin
        trig[trig <= thres] = thres
        return trig

    @staticmethod
    def _box(b1, b2, channel, image_size, vmin, val):
        shape = (image_size, image_size) if isinstance(image_size, int) else tuple(image_size)
        trig = torch.full(size=(channel, *shape), fill_value=vmin)
        trig[:, b1[0]:b2[0], b1[1]:b2[1]] = val
        return trig

    @staticmethod
    def _box_coord(x, y):
        if x < 0 or y < 0:
            raise ValueError("Argument x, y should > 0")
        return (-(y + Backdoor.TRIGGER_GAP_Y), -(x + Backdoor.TRIGGER_GAP_X)), (-Backdoor.TRIGGER_GAP_Y, -Backdoor.TRIGGER_GAP_X)

    def get_trigger(self, type: str, channel: int, image_size: int, vmin=DEFAULT_VMIN, vmax=DEFAULT_VMAX) -> torch.Tensor:
        # dataset.py:526-597
        if type in self._WHITE or type in self._GREY:
            k = self._WHITE.get(type) or self._GREY.get(type)
            b1, b2 = self._box_coord(k, k)
            val = vmax if type in self._WHITE else (vmin + vmax) / 2
            return self._box(b1, b2, channel, image_size, vmin, val)
        if type == self.TRIGGER_GLASSES:
            return self._img_trigger(self.GLASSES_IMG, image_size, channel, int(image_size * 0.625), vmin, vmax)
        if type in self._STOP:
            return self._img_trigger(self.STOP_SIGN_IMG, image_size, channel, self._STOP[type], vmin, vmax, x=-2, y=-2)
        if type == self.TRIGGER_NONE:
            return torch.full(size=(channel, image_size, image_size), fill_value=vmin)
        if type in ("FASHION", "FASHION_EZ", "MNIST", "MNIST_EZ"):
            raise NotImplementedError(f"trigger {type} needs the torchvision FashionMNIST/MNIST download (no network here)")
        raise ValueError(f"Trigger type {type} isn't found")

    def get_target(self, type: str, trigger: torch.Tensor = None, dx: int = -5, dy: int = -3, vmin=DEFAULT_VMIN,
                   vmax=DEFAULT_VMAX) -> torch.Tensor:
        # dataset.py:627-655
        channel = trigger.shape[0]
        image_size = list(trigger.shape[-2:])
        if type == self.TARGET_TG:
            return self._bg2grey(trigger.clone().detach(), vmin, vmax)
        if type == self.TARGET_SHIFT:
            return self._bg2grey(torch.roll(trigger.clone().detach(), shifts=(0, dy, dx), dims=(0, 1, 2)), vmin, vmax)
        if type == self.TARGET_CORNER:
            return self._bg2grey(self._box((None, None), (10, 10), channel, image_size, vmin, (vmin + vmax) / 2), vmin, vmax)
        if type == self.TARGET_HAT:
            return self._bg2grey(self._load_img(self._asset(self.HAT_IMG), channel, image_size, vmin, vmax), vmin, vmax)
        if type == self.TARGET_CAT:
            return self._bg2grey(self._load_img(self._asset(self.CAT_IMG), channel, image_size, vmin, vmax), vmin, vmax)
        if type == self.TARGET_SHOE:
            raise NotImplementedError("target SHOE needs the FashionMNIST download (no network here)")
        raise NotImplementedError(f"Target type {type} isn't found")


class DatasetLoader:
    """Poisoned dataset with the reference's builder API (dataset.py:48-380):
        DatasetLoader(root, name, batch_size).set_poison(trigger_type, target_type, clean_rate, poison_rate)
            .prepare_dataset(mode)  ->  .get_dataloader() / .device_batches()
    """
    MODE_FIXED = "FIXED"
    MODE_FLEX = "FLEX"
    MNIST, CIFAR10, CELEBA, LSUN_CHURCH, LSUN_BEDROOM, CELEBA_HQ = "MNIST", "CIFAR10", "CELEBA", "LSUN-CHURCH", "LSUN-BEDROOM", "CELEBA-HQ"
    TRAIN, TEST = "train", "test"
    PIXEL_VALUES, TARGET, IS_CLEAN, IMAGE, LABEL = "pixel_values", "target", "is_clean", "image", "label"
    _SIZES = {"MNIST": (1, 32, 70000), "CIFAR10": (3, 32, 60000), "CELEBA": (3, 64, 202599), "CELEBA-HQ": (3, 256, 30000),
              "LSUN-CHURCH": (3, 256, 126227), "LSUN-BEDROOM": (3, 256, 303125)}

    def __init__(self, root: str, name: str, label: int = None, channel: int = None, image_size: int = None,
                 batch_size: int = 512, seed: int = 0, num_images: int = None, device=None):
        if name not in self._SIZES:
            raise NotImplementedError(f"Undefined dataset: {name}")
        self._root, self._name, self._batch_size, self._seed = root, name, batch_size, seed
        self._label, self._subset = label, None
        c, s, n = self._SIZES[name]
        self._channel = channel or c
        self._image_size = image_size or s
        self._vmin, self._vmax = DEFAULT_VMIN, DEFAULT_VMAX
        self._device = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        path = None
        for ext in (".npy", ".npz"):
            cand = os.path.join(root or ".", name.lower() + "_u8" + ext)
            if os.path.exists(cand):
                path = cand
        if path is not None:
            arr = np.load(path)
            arr = arr[arr.files[0]] if hasattr(arr, "files") else arr
            self._images = torch.from_numpy(np.ascontiguousarray(arr)).to(torch.uint8)
            self.source = f"file:{path}"
        else:
            n = num_images or n
            g = torch.Generator().manual_seed(seed)
            self._images = torch.randint(0, 256, (n, self._image_size, self._image_size, self._channel), generator=g,
                                         dtype=torch.uint8)
            self.source = "synthetic"
        self._labels = torch.full((len(self._images),), -1.0)
        self._backdoor = Backdoor(root=root)
        self.set_poison(Backdoor.TRIGGER_NONE, Backdoor.TARGET_TG, clean_rate=1.0, poison_rate=0.0)
        self._is_poison = torch.zeros(len(self._images), dtype=torch.bool)
        self._dev_images = None

    def set_poison(self, trigger_type: str, target_type: str, target_dx: int = -5, target_dy: int = -3,
                   clean_rate: float = 1.0, poison_rate: float = 0.2):
        if self._root is not None:
            self._backdoor = Backdoor(root=self._root)
        self._trigger_type, self._target_type = trigger_type, target_type
        self._trigger = self._backdoor.get_trigger(type=trigger_type, channel=self._channel, image_size=self._image_size,
                                                   vmin=self._vmin, vmax=self._vmax)
        self._target = self._backdoor.get_target(type=target_type, trigger=self._trigger, dx=target_dx, dy=target_dy)
        self._clean_rate, self._poison_rate = clean_rate, poison_rate
        return self

    def prepare_dataset(self, mode: str = "FIXED"):
        """FIXED (dataset.py:162-201): backdoor_n = int(N * poison_rate) of the N samples become backdoor samples,
        clean_rate is ignored.  FLEX (dataset.py:225-243): train_n = int(N * clean_rate) clean samples PLUS
        test_n = int(N * poison_rate) backdoor samples, disjoint, so the training set has train_n + test_n rows.
        The reference picks them with an UNSEEDED train_test_split (SURVEY D-5); here the choice is a seeded
        permutation so runs are reproducible.  The class filter (`label`, dataset.py:247-248) needs real labels."""
        if self._label is not None:
            if bool((self._labels < 0).all()):
                raise NotImplementedError("DatasetLoader(label=...) needs a dataset with labels (synthetic / label-free arrays have none)")
            keep = torch.isin(self._labels.long(), torch.as_tensor(list(self._label) if hasattr(self._label, "__iter__") else [self._label]))
            self._images, self._labels = self._images[keep], self._labels[keep]
        n = len(self._images)
        perm = torch.randperm(n, generator=torch.Generator().manual_seed(self._seed))
        self._is_poison = torch.zeros(n, dtype=torch.bool)
        if mode == self.MODE_FIXED:
            if float(self._poison_rate) < 0 or float(self._poison_rate) > 1:
                raise ValueError(f"In {self.MODE_FIXED}, poison rate should <= 1.0 and >= 0.0")
            backdoor_n = int(n * float(self._poison_rate))
            self._is_poison[perm[:backdoor_n]] = True
            self._subset = None
        elif mode == self.MODE_FLEX:
            train_n, test_n = int(n * float(self._clean_rate)), int(n * float(self._poison_rate))
            if train_n + test_n > n:
                raise ValueError(f"In {self.MODE_FLEX}, clean_rate + poison_rate must be <= 1 (train_test_split of one dataset)")
            self._is_poison[perm[train_n: train_n + test_n]] = True
            self._subset = perm[: train_n + test_n].sort().values      # the rows that take part in training
        else:
            raise NotImplementedError(f"Argument mode: {mode} isn't defined")
        self._dev_images = None
        return self

    def _rows(self):
        """indices (into the resident image array) of the prepared dataset"""
        return torch.arange(len(self._images)) if getattr(self, "_subset", None) is None else self._subset

    # ---- device-resident batches (the product path) --------------------------------------------------------
    def to_device(self, device=None):
        device = torch.device(device) if device is not None else self._device
        self._device = device
        self._dev_images = self._images.to(device)
        self._dev_poison = self._is_poison.to(device)
        self._dev_trigger = self._trigger.to(device)
        self._dev_target = self._target.to(device)
        return self

    def device_batch_rows(self, shuffle=True, epoch=0, rank=0, world=1, flip=True):
        """The product path: yield (rows int64 [B], flip uint8 [B] or None, is_poison [B]) on the device -- row numbers
        into the HBM-resident uint8 array (`self.device_images`) for the fused gather + flip + normalize + blend +
        q_sample kernel (bd_poison_qsample row_index / flip); no image bytes move before that kernel.  Same
        permutation, rank slicing and flip draws as device_batches()."""
        if self._dev_images is None:
            self.to_device()
        rows = self._rows()
        n = len(rows)
        g = torch.Generator().manual_seed(self._seed * 100003 + epoch)
        idx = rows[torch.randperm(n, generator=g)] if shuffle else rows
        gb = self._batch_size * world
        for s in range(0, n, gb):
            chunk = idx[s: s + gb]
            if world > 1 and len(chunk) % world:
                chunk = torch.cat([chunk, idx[: world - len(chunk) % world]])
            sel = chunk[rank::world].to(self._device)
            f = (torch.rand(len(chunk), generator=g)[rank::world] < 0.5).to(torch.uint8).to(self._device) if flip else None
            yield sel, f, self._dev_poison[sel]

    @property
    def device_images(self):
        if self._dev_images is None:
            self.to_device()
        return self._dev_images

    def device_batches(self, shuffle=True, epoch=0, rank=0, world=1, flip=True):
        """Yield (images_u8 [B,H,W,C], is_poison [B]) already on the device; rank r takes every world-th row of each
        global batch.  Every rank gets the SAME number of rows in every step (the last, partial global batch is
        wrap-padded with rows from the start of the permutation, like DistributedSampler): unequal or empty per-rank
        batches would mis-weight the 1/world gradient mean or hang the all-reduce."""
        if self._dev_images is None:
            self.to_device()
        rows = self._rows()
        n = len(rows)
        g = torch.Generator().manual_seed(self._seed * 100003 + epoch)
        idx = rows[torch.randperm(n, generator=g)] if shuffle else rows
        gb = self._batch_size * world
        for s in range(0, n, gb):
            chunk = idx[s: s + gb]
            if world > 1 and len(chunk) % world:
                chunk = torch.cat([chunk, idx[: world - len(chunk) % world]])
            sel = chunk[rank::world].to(self._device)
            img = self._dev_images[sel]
            if flip:   # RandomHorizontalFlip(p=0.5) (dataset.py:127-128), on the device
                f = torch.rand(len(chunk), generator=g)[rank::world] < 0.5
                img = torch.where(f.to(self._device)[:, None, None, None], img.flip(2), img)
            yield img, self._dev_poison[sel]

    # ---- reference-style dict batches -----------------------------------------------------------------------
    def _transform(self, u8):
        x = u8.permute(0, 3, 1, 2).float() / 255.0
        return normalize(x, vmin_in=0.0, vmax_in=1.0, vmin_out=self._vmin, vmax_out=self._vmax)

    def get_mask(self, trigger: torch.Tensor) -> torch.Tensor:
        return torch.where(trigger > self._vmin, 0, 1)

    def get_dataloader(self, shuffle=True):
        """The reference's dict batches (dataset.py:43-47, 288-315).  normalize / mask / blend run in the fused
        device kernel (bd_poison_qsample emits R, x0 and the normalised image); batches stay on the device."""
        from . import ops
        if self._dev_images is None:
            self.to_device()
        dev = self._device
        if dev.type != "cuda":
            raise RuntimeError("DatasetLoader.get_dataloader needs a GPU: the per-sample transforms run in libbd_hip.so")
        rows = self._rows()
        n = len(rows)
        idx = rows[torch.randperm(n, generator=torch.Generator().manual_seed(self._seed))] if shuffle else rows
        a = torch.ones(1, device=dev) * 0.5
        for s in range(0, n, self._batch_size):
            sel = idx[s: s + self._batch_size].to(dev)
            u8 = self._dev_images[sel]
            pois = self._dev_poison[sel]
            B = len(sel)
            zeros = torch.zeros(B, self._channel, self._image_size, self._image_size, device=dev)
            tz = torch.zeros(B, dtype=torch.int64, device=dev)
            _, _, R, x0, img = ops.poison_qsample(u8, pois, self._dev_trigger, self._dev_target, zeros, tz, a, a,
                                                  vmin=self._vmin, want_batch=True, want_image=True)
            yield {self.PIXEL_VALUES: R, self.TARGET: x0, self.IMAGE: img, self.LABEL: self._labels[sel.cpu()].to(dev),
                   self.IS_CLEAN: ~pois}

    def get_dataset(self):
        return self

    def __len__(self):
        return len(self._rows())

    def __getitem__(self, i):
        i = int(self._rows()[i])
        img = self._transform(self._images[i: i + 1])[0]
        return {self.IMAGE: img, self.LABEL: self._labels[i], self.IS_CLEAN: not bool(self._is_poison[i])}

    @property
    def len(self):
        return len(self)

    @property
    def num_batch(self):
        return (len(self) + self._batch_size - 1) // self._batch_size

    @property
    def trigger(self):
        return self._trigger

    @property
    def target(self):
        return self._target

    @property
    def name(self):
        return self._name

    @property
    def root(self):
        return self._root

    @property
    def batch_size(self):
        return self._batch_size

    @property
    def channel(self):
        return self._channel

    @property
    def image_size(self):
        return self._image_size
