"""Image pre/post-processing of the Fill pipeline, restating VaeImageProcessor (reference:
diffusers/src/diffusers/image_processor.py: preprocess :587-716, postprocess :718-771, binarize :523-538, denormalize
:227-239, pt_to_numpy :196-209, numpy_to_pil :133-154).

Two layers: `to_raw()` is the host part that has to stay on the host (PIL decoding / resizing, list handling) and returns
the still un-normalised pixels -- uint8 for PIL inputs --; the arithmetic (value / 255, 2x - 1, binarise, image * (1 - mask),
the bf16 cast and the NHWC layout) then runs on the device in `tfx_prep_image` / `tfx_pack_mask` (pipeline.py).
`preprocess()` / `postprocess()` keep the reference's all-host tensor interface (used by callers that want the tensors, and
as the CPU-testable statement of the same arithmetic)."""
from __future__ import annotations

from typing import List, Optional, Union

import numpy as np
import PIL.Image
import torch
import torch.nn.functional as F

_RESAMPLE = {"lanczos": PIL.Image.LANCZOS, "bilinear": PIL.Image.BILINEAR, "bicubic": PIL.Image.BICUBIC,
             "nearest": PIL.Image.NEAREST}


def pil_resample_tables(in_size: int, out_size: int):
    """Window bounds [out, 2] and fixed-point coefficients [out, ksize] (int32, 22 fractional bits) of Pillow's BICUBIC
    resampling of one axis from in_size to out_size -- libImaging/Resample.c: bicubic_filter (a = -0.5, support 2),
    precompute_coeffs, normalize_coeffs_8bpc, evaluated in the same order in C doubles (= Python floats).  PIL.Image.resize's
    default filter for "RGB" / "L" images; consumed by tfx_resample_u8."""
    import math

    def bicubic(x: float) -> float:
        a = -0.5
        x = -x if x < 0.0 else x
        if x < 1.0:
            return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
        if x < 2.0:
            return (((x - 5) * x + 8) * x - 4) * a
        return 0.0

    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << 22)) if k < 0 else int(0.5 + k * (1 << 22))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


class VaeImageProcessor:
    def __init__(self, do_resize: bool = True, vae_scale_factor: int = 8, vae_latent_channels: int = 4,
                 resample: str = "lanczos", do_normalize: bool = True, do_binarize: bool = False,
                 do_convert_rgb: bool = False, do_convert_grayscale: bool = False):
        if do_convert_rgb and do_convert_grayscale:
            raise ValueError("`do_convert_rgb` and `do_convert_grayscale` can not both be set to `True`")
        self.do_resize, self.vae_scale_factor, self.vae_latent_channels = do_resize, vae_scale_factor, vae_latent_channels
        self.resample, self.do_normalize, self.do_binarize = resample, do_normalize, do_binarize
        self.do_convert_rgb, self.do_convert_grayscale = do_convert_rgb, do_convert_grayscale

    # -- helpers -------------------------------------------------------------------------------------------
    def get_default_height_width(self, image, height=None, width=None):
        if height is None:
            height = image.height if isinstance(image, PIL.Image.Image) else image.shape[2 if isinstance(image, torch.Tensor) else 1]
        if width is None:
            width = image.width if isinstance(image, PIL.Image.Image) else image.shape[3 if isinstance(image, torch.Tensor) else 2]
        return height - height % self.vae_scale_factor, width - width % self.vae_scale_factor

    def _resize(self, image, height, width):
        if isinstance(image, PIL.Image.Image):
            return image.resize((width, height), resample=_RESAMPLE[self.resample])
        return F.interpolate(image, size=(height, width))

    @staticmethod
    def pil_to_numpy(images: List[PIL.Image.Image]) -> np.ndarray:
        arr = [np.array(i).astype(np.float32) / 255.0 for i in images]
        arr = [a[..., None] if a.ndim == 2 else a for a in arr]
        return np.stack(arr, axis=0)

    @staticmethod
    def numpy_to_pt(images: np.ndarray) -> torch.Tensor:
        if images.ndim == 3:
            images = images[..., None]
        return torch.from_numpy(images.transpose(0, 3, 1, 2))

    @staticmethod
    def binarize(image: torch.Tensor) -> torch.Tensor:
        image[image < 0.5] = 0
        image[image >= 0.5] = 1
        return image

    # -- host half of the device path ------------------------------------------------------------------------
    def to_raw(self, image, height: Optional[int] = None, width: Optional[int] = None) -> torch.Tensor:
        """Same input handling as `preprocess` (formats, lists, resize, RGB / grayscale conversion) WITHOUT the arithmetic:
        returns uint8 [B, H, W, C] ([B, H, W] for grayscale) for PIL inputs, float32 [B, C, H, W] for numpy / torch
        inputs.  Normalisation / binarisation are left to the device kernels."""
        if isinstance(image, torch.Tensor) and image.dtype == torch.uint8:
            # already-decoded pixels, the layout the PIL branch below produces ([B, H, W, 3], or [B, H, W] for a grey mask) --
            # e.g. a canvas composed on the device (ops.compose_canvas); no resize on this path
            want = 3 if self.do_convert_grayscale else 4
            if image.ndim != want or (want == 4 and image.shape[-1] != 3):
                raise ValueError(f"uint8 tensor input must be [B, H, W{', 3' if want == 4 else ''}], got {tuple(image.shape)}")
            if height is not None and width is not None and tuple(image.shape[1:3]) != (height, width):
                raise ValueError(f"uint8 tensor input is {tuple(image.shape[1:3])}, the call asks for {(height, width)}: resize before composing")
            return image
        if self.do_convert_grayscale and isinstance(image, (torch.Tensor, np.ndarray)) and image.ndim == 3:
            if isinstance(image, torch.Tensor):
                image = image.unsqueeze(1)
            else:
                image = np.expand_dims(image, axis=0 if image.shape[-1] == 1 else -1)
        if not isinstance(image, list):
            image = [image]
        first = image[0]
        if isinstance(first, PIL.Image.Image):
            if self.do_resize:
                height, width = self.get_default_height_width(first, height, width)
                image = [self._resize(i, height, width) for i in image]
            if self.do_convert_rgb:
                image = [i.convert("RGB") for i in image]
            elif self.do_convert_grayscale:
                image = [i.convert("L") for i in image]
            arr = [np.array(i) for i in image]
            if any(a.dtype != np.uint8 for a in arr):   # 16-bit / float PIL modes: the all-host arithmetic
                return self.numpy_to_pt(self.pil_to_numpy(image))
            return torch.from_numpy(np.stack(arr, axis=0))
        if isinstance(first, np.ndarray):
            image = np.concatenate(image, axis=0) if first.ndim == 4 else np.stack(image, axis=0)
            image = self.numpy_to_pt(image)
        elif isinstance(first, torch.Tensor):
            image = torch.cat(image, dim=0) if first.ndim == 4 else torch.stack(image, dim=0)
            if self.do_convert_grayscale and image.ndim == 3:
                image = image.unsqueeze(1)
        else:
            raise ValueError("Input is in incorrect format. Currently, we only support PIL.Image.Image, np.ndarray, torch.Tensor")
        height, width = self.get_default_height_width(image, height, width)
        if self.do_resize and tuple(image.shape[-2:]) != (height, width):
            image = self._resize(image, height, width)
        return image.float()

    # -- preprocess ----------------------------------------------------------------------------------------
    def preprocess(self, image, height: Optional[int] = None, width: Optional[int] = None) -> torch.Tensor:
        if self.do_convert_grayscale and isinstance(image, (torch.Tensor, np.ndarray)) and image.ndim == 3:
            if isinstance(image, torch.Tensor):
                image = image.unsqueeze(1)
            else:
                image = np.expand_dims(image, axis=0 if image.shape[-1] == 1 else -1)
        if not isinstance(image, list):
            image = [image]
        first = image[0]
        if isinstance(first, PIL.Image.Image):
            if self.do_resize:
                height, width = self.get_default_height_width(first, height, width)
                image = [self._resize(i, height, width) for i in image]
            if self.do_convert_rgb:
                image = [i.convert("RGB") for i in image]
            elif self.do_convert_grayscale:
                image = [i.convert("L") for i in image]
            image = self.numpy_to_pt(self.pil_to_numpy(image))
        elif isinstance(first, np.ndarray):
            image = np.concatenate(image, axis=0) if first.ndim == 4 else np.stack(image, axis=0)
            image = self.numpy_to_pt(image)
            height, width = self.get_default_height_width(image, height, width)
            if self.do_resize:
                image = self._resize(image, height, width)
        elif isinstance(first, torch.Tensor):
            image = torch.cat(image, dim=0) if first.ndim == 4 else torch.stack(image, dim=0)
            if self.do_convert_grayscale and image.ndim == 3:
                image = image.unsqueeze(1)
            if image.shape[1] == self.vae_latent_channels:
                return image
            height, width = self.get_default_height_width(image, height, width)
            if self.do_resize:
                image = self._resize(image, height, width)
        else:
            raise ValueError("Input is in incorrect format. Currently, we only support PIL.Image.Image, np.ndarray, torch.Tensor")
        do_normalize = self.do_normalize
        if do_normalize and image.min() < 0:
            do_normalize = False  # already in [-1, 1] (deprecated input convention)
        if do_normalize:
            image = 2.0 * image - 1.0
        if self.do_binarize:
            image = self.binarize(image)
        return image

    # -- postprocess ---------------------------------------------------------------------------------------
    def postprocess(self, image: torch.Tensor, output_type: str = "pil"):
        if output_type == "latent":
            return image
        image = (image / 2 + 0.5).clamp(0, 1) if self.do_normalize else image
        if output_type == "pt":
            return image
        arr = image.cpu().permute(0, 2, 3, 1).float().numpy()
        if output_type == "np":
            return arr
        if output_type == "pil":
            u8 = (arr * 255).round().astype("uint8")
            if u8.shape[-1] == 1:
                return [PIL.Image.fromarray(a.squeeze(), mode="L") for a in u8]
            return [PIL.Image.fromarray(a) for a in u8]
        raise ValueError(f"unknown output_type {output_type}")
