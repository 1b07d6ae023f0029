"""Caller-side logic of TextFlux: prompt strings, glyph rendering, scene+glyph concatenation, result crop.

Own counterpart of the reference harness (run_inference.py:19-40 prompts, :118-181 single-line strip, :217-376 multi-line
render, :378-384 concat direction, :409-467 concat + crop).  Geometry and strings are pinned by the known answers of
SURVEY.md Appendix F (tests/test_host_logic.py); glyph *pixels* are not (the reference's TTF and cv2 are absent here:
region detection uses scipy.ndimage connected components + axis-aligned boxes instead of cv2.minAreaRect).
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import numpy as np
from PIL import Image, ImageDraw, ImageFont

TEXT_HEIGHT_RATIO = 0.15625  # run_inference.py:164
DEFAULT_FONT = "resource/font/Arial-Unicode-Regular.ttf"

PROMPT_TEMPLATE2 = (
    "The pair of images highlights some white words on a black background, as well as their style on a real-world "
    "scene image. [IMAGE1] is a template image rendering the text, with the words; [IMAGE2] shows the text content "
    "naturally and correspondingly integrated into the image."
)


def read_words_from_text(input_text: str) -> List[str]:
    if isinstance(input_text, str) and os.path.exists(input_text):
        with open(input_text, "r", encoding="utf-8") as f:
            return [ln.strip() for ln in f if ln.strip()]
    return [ln.strip() for ln in input_text.splitlines() if ln.strip()]


def generate_prompt(words: Sequence[str]) -> str:
    w = ", ".join(f"'{x}'" for x in words)
    return (
        "The pair of images highlights some white words on a black background, as well as their style on a real-world "
        f"scene image. [IMAGE1] is a template image rendering the text, with the words {w}; [IMAGE2] shows the text "
        f"content {w} naturally and correspondingly integrated into the image."
    )


def load_font(path: Optional[str] = None, size: int = 60):
    try:
        return ImageFont.truetype(path or DEFAULT_FONT, size)
    except (IOError, OSError):
        return ImageFont.load_default()


def draw_glyph(font, text: str, width: int, height: int, max_font_size: int = 140) -> Image.Image:
    """White text centred on a black canvas; font size scaled from a size-50 probe so the text fills 90 % of the box,
    capped at 140 (200 for canvases wider than 1280), floor 10."""
    img = Image.new("RGB", (width, height), "black")
    if not text or not text.strip():
        return img
    probe = 50

    def variant(sz):
        try:
            return font.font_variant(size=sz)
        except Exception:
            return font

    l, t, r, b = variant(probe).getbbox(text)
    ratio = min(width * 0.9 / max(r - l, 1), height * 0.9 / max(b - t, 1))
    cap = 200 if width > 1280 else max_font_size
    size = max(min(int(probe * ratio), cap), 10)
    ImageDraw.Draw(img).text((width / 2, height / 2), text, font=variant(size), fill="white", anchor="mm")
    return img


def render_single_line(scene: Image.Image, words: Sequence[str], font_path: Optional[str] = None):
    w, _ = scene.size
    strip_h = int(w * TEXT_HEIGHT_RATIO)
    return draw_glyph(load_font(font_path), " ".join(words), w, strip_h), strip_h


def mask_regions(mask: Image.Image, min_area: int = 16) -> List[Tuple[int, int, int, int]]:
    """Bounding boxes (l, t, r, b) of the white regions of the mask, reading order (top-to-bottom, left-to-right)."""
    from scipy import ndimage
    m = np.array(mask.convert("L")) > 127
    lab, n = ndimage.label(m)
    boxes = []
    for sl in ndimage.find_objects(lab):
        if sl is None:
            continue
        t, b, l, r = sl[0].start, sl[0].stop, sl[1].start, sl[1].stop
        if (b - t) * (r - l) >= min_area:
            boxes.append((l, t, r, b))
    boxes.sort(key=lambda bx: (bx[1], bx[0]))
    return boxes


def render_multiline(scene: Image.Image, mask: Image.Image, texts: Sequence[str], font_path: Optional[str] = None):
    """One text line per mask region, drawn at the region's position on a black canvas of the scene's size."""
    canvas = Image.new("RGB", scene.size, "black")
    font = load_font(font_path)
    for (l, t, r, b), text in zip(mask_regions(mask), texts):
        canvas.paste(draw_glyph(font, text, max(r - l, 1), max(b - t, 1)), (l, t))
    return canvas


def choose_concat_direction(height: int, width: int) -> str:
    return "horizontal" if height > width else "vertical"


def compose(scene: Image.Image, mask: Image.Image, words: Sequence[str], font_path: Optional[str] = None):
    """-> (combined image, combined mask, meta).  Glyph image goes first (top / left); its mask is black."""
    scene, mask = scene.convert("RGB"), mask.convert("RGB")
    if len(words) > 1:
        glyph = render_multiline(scene, mask, words, font_path)
        direction = choose_concat_direction(scene.size[1], scene.size[0])
        stack = np.hstack if direction == "horizontal" else np.vstack
        meta = dict(mode="multiline", direction=direction)
    else:
        glyph, strip_h = render_single_line(scene, words, font_path)
        stack, meta = np.vstack, dict(mode="singleline", direction="vertical", strip=strip_h, orig_h=scene.size[1])
    black = Image.new("RGB", glyph.size, "black")
    image = Image.fromarray(stack((np.array(glyph), np.array(scene))))
    cmask = Image.fromarray(stack((np.array(black), np.array(mask))))
    return image, cmask, meta


def pipe_size(image: Image.Image) -> Tuple[int, int]:
    """(width, height) floored to multiples of 32 (run_inference.py:65-69)."""
    w, h = image.size
    return (w // 32) * 32, (h // 32) * 32


def crop_box(result_size: Tuple[int, int], meta: dict) -> Tuple[int, int, int, int]:
    w, h = result_size
    if meta["mode"] == "multiline":
        return (w // 2, 0, w, h) if meta["direction"] == "horizontal" else (0, h // 2, w, h)
    top = int(h * (meta["strip"] / (meta["orig_h"] + meta["strip"])))
    return (0, top, w, h)


def synthetic_case(width: int, height: int, multiline: bool = False, seed: int = 0):
    """Deterministic scene / mask / words for benchmarks and tests (no assets needed)."""
    rng = np.random.default_rng(seed)
    scene = Image.fromarray(rng.integers(0, 256, (height, width, 3), dtype=np.uint8))
    m = np.zeros((height, width), np.uint8)
    if multiline:
        m[height // 6: height // 3, width // 8: 7 * width // 8] = 255
        m[height // 2: 2 * height // 3, width // 4: 3 * width // 4] = 255
        words = ["HELLO", "WORLD"]
    else:
        m[height // 3: 2 * height // 3, width // 8: 7 * width // 8] = 255
        words = ["TEXTFLUX"]
    return scene, Image.fromarray(m).convert("RGB"), words
