"""Image-space upscaler hand-off used by the non-latent hires fix.

The engine's part is decode -> [host image resize] -> encode; the resize itself stays the reference's job.  This module is
the seam: the registry ``shared.sd_upscalers`` of ``UpscalerData(name, path, scaler)`` entries the reference's code looks names
up in (modules/images.py:276, modules/modelloader.py:136), the ``Upscaler.upscale`` driver loop (modules/upscaler.py:54-76) and
the three built-in PIL scalers (None / Lanczos / Nearest, modules/upscaler.py:107-154).  Model upscalers (ESRGAN, SwinIR, ...)
register their own ``UpscalerData`` whose ``scaler.upscale(img, scale, path)`` is called as in the reference.
"""
from __future__ import annotations

from PIL import Image

from . import shared

LANCZOS = (Image.Resampling.LANCZOS if hasattr(Image, 'Resampling') else Image.LANCZOS)
NEAREST = (Image.Resampling.NEAREST if hasattr(Image, 'Resampling') else Image.NEAREST)


class UpscalerData:
    """A selectable upscaler: what shared.sd_upscalers holds (modules/upscaler.py:87-104)."""

    def __init__(self, name, path, upscaler=None, scale=4, model=None):
        self.name, self.data_path, self.local_data_path = name, path, path
        self.scaler, self.scale, self.model = upscaler, scale, model

    def __repr__(self):
        return f"<UpscalerData name={self.name} path={self.data_path} scale={self.scale}>"


class Upscaler:
    """Base class with the reference's driver (modules/upscaler.py:54-76): a scaler may enlarge by less than asked for, so it
    is applied up to three times until the (8-aligned) destination size is covered, then the result is fitted with LANCZOS."""
    name = None

    def __init__(self):
        self.scale = 1
        self.scalers = []

    def do_upscale(self, img, selected_model=None):
        return img

    def upscale(self, img, scale, selected_model=None):
        self.scale = scale
        dest_w, dest_h = int((img.width * scale) // 8 * 8), int((img.height * scale) // 8 * 8)
        for attempt in range(3):
            covered = img.width >= dest_w and img.height >= dest_h
            if (covered and (attempt > 0 or scale != 1)) or shared.state.interrupted:
                break
            before = (img.width, img.height)
            img = self.do_upscale(img, selected_model)
            if before == (img.width, img.height):
                break
        if (img.width, img.height) != (dest_w, dest_h):
            img = img.resize((dest_w, dest_h), resample=LANCZOS)
        return img


class _PilUpscaler(Upscaler):
    """The built-in scalers that are one PIL resize (modules/upscaler.py:107-154): "None" (identity), "Lanczos", "Nearest"."""
    resample = None

    def __init__(self):
        super().__init__()
        self.scalers = [UpscalerData(self.name, None, self)]

    def do_upscale(self, img, selected_model=None):
        if self.resample is None:
            return img
        return img.resize((int(img.width * self.scale), int(img.height * self.scale)), resample=self.resample)


class UpscalerNone(_PilUpscaler):
    name = "None"


class UpscalerLanczos(_PilUpscaler):
    name, resample = "Lanczos", LANCZOS


class UpscalerNearest(_PilUpscaler):
    name, resample = "Nearest", NEAREST


def builtin_upscalers():
    """The order modelloader.load_upscalers leaves the built-ins in (:136-141: "None" first, then by name)."""
    return [*UpscalerNone().scalers, *UpscalerLanczos().scalers, *UpscalerNearest().scalers]


def resize_image(resize_mode, im, width, height, upscaler_name=None):
    """modules/images.py:252-291, resize_mode 0 (plain resize to width x height; the hires fix uses no other mode)."""
    if resize_mode != 0:
        raise NotImplementedError("resize modes 1 / 2 (crop / fill) belong to the img2img front-end")
    upscaler_name = upscaler_name or getattr(shared.opts, "upscaler_for_img2img", None)
    if upscaler_name is None or upscaler_name == "None" or im.mode == 'L':
        return im.resize((width, height), resample=LANCZOS)
    scale = max(width / im.width, height / im.height)
    if scale > 1.0:
        upscalers = [x for x in shared.sd_upscalers if x.name == upscaler_name]
        upscaler = upscalers[0] if upscalers else shared.sd_upscalers[0]
        im = upscaler.scaler.upscale(im, scale, upscaler.data_path)
    if im.width != width or im.height != height:
        im = im.resize((width, height), resample=LANCZOS)
    return im
