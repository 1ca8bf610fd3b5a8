"""Image-space mask helpers of the img2img / inpainting front-end (host side, PIL).

Mirrors the interface of the reference's ``modules/masking.py`` (get_crop_region_v2 :4-19, get_crop_region :22-36, expand_crop_region
:39-78, fill :81-96) plus the mask / overlay helpers ``modules/processing.py`` keeps next to them (create_binary_mask :90-98, uncrop
:70-77, apply_overlay :80-93) and the separable Gaussian mask blur of ``StableDiffusionProcessingImg2Img.init`` (:1621-1631, there through
cv2.GaussianBlur; cv2 is not a dependency of this package: the same kernel is applied in numpy).  Pinned by
tests/golden/img2img_frontend.npz (fixtures from the reference's own files) except the blur, whose reference needs cv2.
"""
from __future__ import annotations

import numpy as np
from PIL import Image, ImageFilter, ImageOps


def get_crop_region_v2(mask, pad=0):
    """Bounding box (x1, y1, x2, y2) of the non-zero area of an L-mode mask (PIL image or 2-D array), grown by ``pad`` and clipped to
    the mask; None for an all-black mask."""
    if not isinstance(mask, Image.Image):
        mask = Image.fromarray(mask)
    box = mask.getbbox()
    if box is None:
        return None
    if not pad:
        return box
    w, h = mask.size
    return max(box[0] - pad, 0), max(box[1] - pad, 0), min(box[2] + pad, w), min(box[3] + pad, h)


def get_crop_region(mask, pad=0):
    """The pre-1.9 form: an all-black mask yields the (possibly inverted) box (w - pad, h - pad, pad, pad) clipped to the mask."""
    if not isinstance(mask, Image.Image):
        mask = Image.fromarray(mask)
    box = get_crop_region_v2(mask, pad)
    if box:
        return box
    w, h = mask.size
    return max(w - pad, 0), max(h - pad, 0), min(pad, w), min(pad, h)


def _grow(lo, hi, want, limit):
    """Grow the interval [lo, hi) to ``want`` pixels, split as evenly as integer halving allows (extra pixel at the far end), then
    slide it back inside [0, limit]."""
    extra = int(want - (hi - lo))
    lo -= extra // 2
    hi += extra - extra // 2
    if hi >= limit:
        over = hi - limit
        hi -= over
        lo -= over
    if lo < 0:
        hi -= lo
        lo = 0
    if hi >= limit:
        hi = limit
    return lo, hi


def expand_crop_region(crop_region, processing_width, processing_height, image_width, image_height):
    """Grow a crop region to the aspect ratio the image will be processed at (e.g. a 128x32 mask box processed at 512x512 becomes
    128x128), staying inside the image."""
    x1, y1, x2, y2 = crop_region
    target = processing_width / processing_height
    if (x2 - x1) / (y2 - y1) > target:                       # too wide: grow vertically
        y1, y2 = _grow(y1, y2, (x2 - x1) / target, image_height)
    else:                                                    # too tall: grow horizontally
        x1, x2 = _grow(x1, x2, (y2 - y1) * target, image_width)
    return x1, y1, x2, y2


# (blur radius, number of times the blurred layer is composited) — coarse to fine, as the reference's fill()
_FILL_LADDER = ((256, 1), (64, 1), (16, 2), (4, 4), (2, 2), (0, 1))


def fill(image, mask):
    """"fill" masked content: the masked area is painted with colours bled in from its surroundings — the unmasked pixels
    (premultiplied alpha) are Gaussian-blurred at decreasing radii and composited over each other."""
    size = (image.width, image.height)
    kept = Image.new('RGBa', size)
    kept.paste(image.convert("RGBA").convert("RGBa"), mask=ImageOps.invert(mask.convert('L')))
    kept = kept.convert('RGBa')
    acc = Image.new('RGBA', size)
    for radius, repeats in _FILL_LADDER:
        layer = kept.filter(ImageFilter.GaussianBlur(radius)).convert('RGBA')
        for _ in range(repeats):
            acc.alpha_composite(layer)
    return acc.convert("RGB")


def create_binary_mask(image, round=True):
    """Gradio hands masks over as RGBA: when the alpha channel carries information it IS the mask (thresholded at 128 when ``round``),
    otherwise the luminance is."""
    if image.mode == 'RGBA' and image.getextrema()[-1] != (255, 255):
        alpha = image.split()[-1].convert("L")
        return alpha.point(lambda v: 255 if v > 128 else 0) if round else alpha
    return image.convert('L')


def gaussian_kernel_1d(sigma):
    """The kernel cv2.GaussianBlur(mask, (k, 1), sigma) applies at modules/processing.py:1621-1631: k = 2 * int(2.5 * sigma + 0.5) + 1
    taps of exp(-x^2 / (2 sigma^2)), normalised (cv2.getGaussianKernel for sigma > 0)."""
    k = 2 * int(2.5 * sigma + 0.5) + 1
    x = np.arange(k, dtype=np.float64) - (k - 1) / 2
    w = np.exp(-(x * x) / (2.0 * float(sigma) ** 2))
    return w / w.sum()


def gaussian_blur_axis(np_mask, sigma, axis):
    """One pass of that blur along ``axis`` (1 = x, 0 = y) of a uint8 array, border reflected without repeating the edge pixel
    (cv2.BORDER_REFLECT_101, its default), rounded half to even like cv2's saturate_cast.  cv2 evaluates uint8 images in fixed point:
    results may differ from it by one level."""
    if sigma <= 0:
        return np_mask
    w = gaussian_kernel_1d(sigma)
    r = (len(w) - 1) // 2
    a = np.moveaxis(np.asarray(np_mask, dtype=np.float64), axis, 0)
    n = a.shape[0]
    idx = np.arange(-r, n + r)
    if n > 1:
        period = 2 * (n - 1)
        idx = np.abs(((idx % period) + period) % period)
        idx = np.where(idx >= n, period - idx, idx)
    else:
        idx = np.zeros_like(idx)
    padded = a[idx]
    out = np.zeros_like(a)
    for t in range(len(w)):
        out += w[t] * padded[t:t + n]
    out = np.clip(np.rint(out), 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def blur_mask(image_mask, blur_x, blur_y):
    """modules/processing.py:1621-1631: x pass then y pass on the L-mode mask."""
    if blur_x > 0:
        image_mask = Image.fromarray(gaussian_blur_axis(np.array(image_mask), blur_x, 1))
    if blur_y > 0:
        image_mask = Image.fromarray(gaussian_blur_axis(np.array(image_mask), blur_y, 0))
    return image_mask


def uncrop(image, dest_size, paste_loc):
    """Put an "only masked" result back where its crop came from: resized (cover mode) to the crop's size on a transparent canvas of
    the full image size."""
    from .upscaler import resize_image
    x, y, w, h = paste_loc
    canvas = Image.new('RGBA', dest_size)
    canvas.paste(resize_image(1, image, w, h), (x, y))
    return canvas


def apply_overlay(image, paste_loc, overlay):
    """Composite the unmasked part of the original (``overlay``, RGBA with the mask as transparency) over the generated image; returns
    (composited RGB image, the generated image before compositing)."""
    if overlay is None:
        return image, image.copy()
    if paste_loc is not None:
        image = uncrop(image, (overlay.width, overlay.height), paste_loc)
    generated = image.copy()
    image = image.convert('RGBA')
    image.alpha_composite(overlay)
    return image.convert('RGB'), generated
