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


def _resize_to(im, w, h, upscaler_name):
    """The inner `resize` of modules/images.py:269-288: Lanczos for masks / no upscaler, else the named upscaler for enlargements."""
    if upscaler_name is None or upscaler_name == "None" or im.mode == 'L':
        return im.resize((w, h), resample=LANCZOS)
    if max(w / im.width, h / im.height) > 1.0:
        named = [x for x in shared.sd_upscalers if x.name == upscaler_name]
        chosen = named[0] if named else shared.sd_upscalers[0]
        im = chosen.scaler.upscale(im, max(w / im.width, h / im.height), chosen.data_path)
    return im if (im.width, im.height) == (w, h) else im.resize((w, h), resample=LANCZOS)


def resize_image(resize_mode, im, width, height, upscaler_name=None):
    """modules/images.py:252-326.  0: stretch to width x height (the hires fix); 1: cover the target keeping the aspect ratio, centred,
    the excess cropped; 2: fit inside the target keeping the aspect ratio, centred, the empty bands filled by stretching the image's
    own border row / column over them."""
    from PIL import Image
    upscaler_name = upscaler_name or getattr(shared.opts, "upscaler_for_img2img", None)
    if resize_mode == 0:
        return _resize_to(im, width, height, upscaler_name)
    ratio, src_ratio = width / height, im.width / im.height
    by_w, by_h = im.width * height // im.height, im.height * width // im.width      # the other side when one side is matched exactly
    if resize_mode == 1:                                     # cover: match the side on which the scaled source would fall short
        src_w, src_h = (width, by_h) if ratio > src_ratio else (by_w, height)
    else:                                                    # fit: match the side on which the scaled source would overflow
        src_w, src_h = (width, by_h) if ratio < src_ratio else (by_w, height)
    resized = _resize_to(im, src_w, src_h, upscaler_name)
    canvas = Image.new("RGB", (width, height))
    x0, y0 = width // 2 - src_w // 2, height // 2 - src_h // 2
    canvas.paste(resized, box=(x0, y0))
    if resize_mode == 2:
        if ratio < src_ratio and y0 > 0:                     # bands above / below: the first / last row stretched over them
            canvas.paste(resized.resize((width, y0), box=(0, 0, width, 0)), box=(0, 0))
            canvas.paste(resized.resize((width, y0), box=(0, resized.height, width, resized.height)), box=(0, y0 + src_h))
        elif ratio > src_ratio and x0 > 0:                   # bands left / right: the first / last column
            canvas.paste(resized.resize((x0, height), box=(0, 0, 0, height)), box=(0, 0))
            canvas.paste(resized.resize((x0, height), box=(resized.width, 0, resized.width, height)), box=(x0 + src_w, 0))
    return canvas
