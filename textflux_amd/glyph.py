"""Caller-side logic of TextFlux: prompt strings, glyph rendering, scene+glyph concatenation, result crop.

Own counterpart of the reference harness (run_inference.py:19-40 prompts, :118-181 single-line strip, :217-376 multi-line
render, :378-384 concat direction, :409-467 concat + crop).  Geometry and strings are pinned by the known answers of
SURVEY.md Appendix F (tests/test_host_logic.py); glyph *pixels* are not (the reference's TTF and cv2 are absent here:
the three OpenCV primitives of the multi-line path are restated on numpy / scipy, see below).
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


def fill_polygon(height: int, width: int, polygon) -> np.ndarray:
    """uint8 [height, width, 3] mask: the polygon (list of [x, y] vertices, truncated to ints as np.int32 does) filled white on
    black, boundary pixels included -- the reference's cv2.fillPoly(zeros, [polygon], (255, 255, 255)) (scripts/run_eval.py:
    92-96).  cv2 is absent here; PIL's scan-line polygon fill plus its outline follows the same rule as OpenCV's FillPoly for
    non-antialiased lines (every edge drawn as an 8-connected Bresenham line, the interior filled by scan line between the
    edges).  Pinned: the axis-aligned rectangle and the triangle of tests/test_batch_driver_cpu.py::
    test_eval_schema_items_follow_the_reference_rule (exact pixel counts), and for slanted / concave polygons the properties both
    rasterisers share (tests/test_host_logic.py::test_fill_polygon_slanted_and_concave_properties: vertices and edge lines
    white, every pixel centre strictly inside white, nothing further than one pixel outside).  Pixel-for-pixel parity with cv2
    on the boundary pixels of slanted edges stays unpinned (no cv2 offline to rasterise goldens with)."""
    pts = [(int(p[0]), int(p[1])) for p in np.asarray(polygon).reshape(-1, 2).tolist()]
    m = Image.new("L", (width, height), 0)
    d = ImageDraw.Draw(m)
    if len(pts) == 1:
        d.point(pts, fill=255)
    elif len(pts) == 2:
        d.line(pts, fill=255)
    elif len(pts) > 2:
        d.polygon(pts, fill=255, outline=255)
    return np.repeat(np.array(m)[:, :, None], 3, axis=2)


def render_single_line(scene: Image.Image, words: Sequence[str], font_path: Optional[str] = None):
    w, _ = scene.size
    strip_h = int(w * TEXT_HEIGHT_RATIO)
    return draw_glyph(load_font(font_path), " ".join(words), w, strip_h), strip_h


# ---------------------------------------------------------------------------------------------------------------------
# Multi-line mode: one text line per mask region, rendered along the region's minimum-area rectangle
# (reference: render_glyph_multi run_inference.py:330-376, draw_glyph2 :217-328).  The reference leans on OpenCV for three
# geometric primitives; cv2 is not available here, so they are restated on numpy / scipy with OpenCV's conventions:
#   findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) + boundingRect  -> 8-connected components of the NON-ZERO mask pixels
#       (nested components inside another component's hole are not separated out -- RETR_EXTERNAL would drop them);
#   minAreaRect  -> rotating calipers over the convex hull of the component's pixels, angle in (0, 90] and (w, h) assigned as
#       OpenCV >= 4.5.1 does (requirements.txt does not pin opencv-python; an axis-aligned 100 x 50 box is ((cx, cy), (50, 100), 90));
#   boxPoints    -> the four corners of that rectangle.
# Everything after the geometry (angle rule, vertical detection, character spacing, font sizing, PIL drawing, rotation,
# compositing) follows the reference line by line.  Pixel parity remains unpinned (no cv2, no TTF here: SURVEY a19).
def mask_regions(mask: Image.Image, min_area: int = 50):
    """[(x, y, w, h, points)] per region, sorted top-to-bottom then left-to-right (:336-344); points = [n, 2] (x, y) pixel
    coordinates of the region."""
    from scipy import ndimage
    m = np.array(mask.convert("L")) != 0
    lab, n = ndimage.label(m, structure=np.ones((3, 3), dtype=bool))
    regions = []
    for idx, sl in enumerate(ndimage.find_objects(lab), start=1):
        if sl is None:
            continue
        y0, y1, x0, x1 = sl[0].start, sl[0].stop, sl[1].start, sl[1].stop
        w, h = x1 - x0, y1 - y0
        if w * h < min_area:
            continue
        ys, xs = np.nonzero(lab[sl] == idx)
        regions.append((x0, y0, w, h, np.stack([xs + x0, ys + y0], axis=1)))
    regions.sort(key=lambda r: (r[1], r[0]))
    return regions


def min_area_rect(points: np.ndarray):
    """((cx, cy), (w, h), angle_deg) of the minimum-area enclosing rectangle, OpenCV >= 4.5.1 conventions."""
    pts = np.unique(np.asarray(points, dtype=np.float64), axis=0)
    if len(pts) >= 3:
        try:
            from scipy.spatial import ConvexHull
            hull = pts[ConvexHull(pts).vertices]
        except Exception:      # degenerate (collinear) point sets
            hull = pts
    else:
        hull = pts
    best = None
    n = len(hull)
    for i in range(n):
        e = hull[(i + 1) % n] - hull[i]
        ln = np.hypot(*e)
        if ln == 0:
            continue
        u = e / ln
        v = np.array([-u[1], u[0]])
        pu, pv = hull @ u, hull @ v
        wu, wv = pu.max() - pu.min(), pv.max() - pv.min()
        if best is None or wu * wv < best[0] - 1e-9:
            c = u * (pu.max() + pu.min()) / 2 + v * (pv.max() + pv.min()) / 2
            best = (wu * wv, u, v, wu, wv, c)
    if best is None:           # a single point / no extent
        c = pts.mean(0)
        return (float(c[0]), float(c[1])), (0.0, 0.0), 90.0
    _, u, v, wu, wv, c = best
    # the side whose direction angle (atan2(dy, dx), y down) falls in (0, 90] carries the width
    cand = []
    for d, ext_d, ext_o in ((u, wu, wv), (-u, wu, wv), (v, wv, wu), (-v, wv, wu)):
        ang = np.degrees(np.arctan2(d[1], d[0]))
        if 0.0 < ang <= 90.0 + 1e-9:
            cand.append((ang, ext_d, ext_o))
    ang, w, h = max(cand) if cand else (90.0, wv, wu)
    return (float(c[0]), float(c[1])), (float(w), float(h)), float(min(ang, 90.0))


def box_points(rect) -> np.ndarray:
    (cx, cy), (w, h), ang = rect
    a = np.radians(ang)
    b, c = np.cos(a) * 0.5, np.sin(a) * 0.5
    p0 = (cx - c * h - b * w, cy + b * h - c * w)
    p1 = (cx + c * h - b * w, cy - b * h - c * w)
    p2 = (2 * cx - p0[0], 2 * cy - p0[1])
    p3 = (2 * cx - p1[0], 2 * cy - p1[1])
    return np.array([p0, p1, p2, p3], dtype=np.float32)


def insert_spaces(text: str, num_spaces: int) -> str:
    return text if len(text) <= 1 else (" " * num_spaces).join(list(text))


def draw_glyph2(font, text: str, polygon: np.ndarray, vertAng: int = 10, scale: float = 1, width: int = 512, height: int = 512,
                add_space: bool = True, scale_factor: int = 2, rotate_resample=Image.BICUBIC,
                downsample_resample=Image.Resampling.LANCZOS) -> np.ndarray:
    """RGBA numpy image [height, width, 4] with `text` rendered along the polygon's minimum-area rectangle (:217-328)."""
    big_w, big_h = width * scale_factor, height * scale_factor
    rect = min_area_rect(polygon * scale_factor * scale)
    box = box_points(rect).astype(np.intp)
    w, h = rect[1]
    angle = rect[2]
    if angle < -45:
        angle += 90
    angle = -angle
    if w < h:
        angle += 90
    vert = False
    if abs(angle) % 90 < vertAng or abs(90 - abs(angle) % 90) % 90 < vertAng:
        _w = max(box[:, 0]) - min(box[:, 0])
        _h = max(box[:, 1]) - min(box[:, 1])
        if _h >= _w:
            vert = True
            angle = 0
    big_img = Image.new("RGBA", (big_w, big_h), (0, 0, 0, 0))
    tmp_draw = ImageDraw.Draw(Image.new("RGB", big_img.size, "white"))
    _, _, _tw, _th = tmp_draw.textbbox((0, 0), text, font=font)
    text_w = 0 if _th == 0 else min(float(w), float(h)) * (_tw / _th)
    if text_w <= max(w, h):
        if len(text) > 1 and not vert and add_space:
            i = 1
            for i in range(1, 100):
                _, _, tw2, th2 = tmp_draw.textbbox((0, 0), insert_spaces(text, i), font=font)
                if th2 != 0 and min(w, h) * (tw2 / th2) > max(w, h):
                    break
            text = insert_spaces(text, i - 1)
        font_size = min(w, h) * 0.80
    else:
        shrink = 0.75 if vert else 0.85
        font_size = min(w, h) / (text_w / max(w, h)) * shrink if text_w != 0 else min(w, h) * 0.80
    try:
        new_font = font.font_variant(size=max(int(font_size), 1))
    except Exception:          # PIL's built-in bitmap font has no variants
        new_font = font
    left, top, right, bottom = new_font.getbbox(text)
    text_width, text_height = right - left, bottom - top
    layer = Image.new("RGBA", big_img.size, (0, 0, 0, 0))
    draw_layer = ImageDraw.Draw(layer)
    cx, cy = rect[0]
    if not vert:
        draw_layer.text((cx - text_width // 2, cy - text_height // 2 - top), text, font=new_font, fill=(255, 255, 255, 255))
    else:
        _w_ = max(box[:, 0]) - min(box[:, 0])
        x_s = min(box[:, 0]) + _w_ // 2 - text_height // 2
        y_s = min(box[:, 1])
        for ch in text:
            draw_layer.text((x_s, y_s), ch, font=new_font, fill=(255, 255, 255, 255))
            _, _t, _, _b = new_font.getbbox(ch)
            y_s += _b
    rotated = layer.rotate(angle, expand=True, center=(cx, cy), resample=rotate_resample)
    xo = int((big_img.width - rotated.width) // 2)
    yo = int((big_img.height - rotated.height) // 2)
    big_img.paste(rotated, (xo, yo), rotated)
    return np.array(big_img.resize((width, height), downsample_resample))


def render_multiline(scene: Image.Image, mask: Image.Image, texts: Sequence[str], font_path: Optional[str] = None):
    """render_glyph_multi (:330-376): region i gets text line i, composited onto a transparent-black canvas."""
    render = Image.new("RGBA", scene.size, (0, 0, 0, 0))
    base_font = load_font(font_path, 40)
    for i, region in enumerate(mask_regions(mask)):
        if i >= len(texts):
            break
        text = texts[i].strip()
        if not text:
            continue
        rgba = draw_glyph2(base_font, text, region[4], vertAng=10, scale=1, width=scene.size[0], height=scene.size[1],
                           add_space=True, scale_factor=1)
        render = Image.alpha_composite(render, Image.fromarray(rgba, mode="RGBA"))
    return render.convert("RGB")


def choose_concat_direction(height: int, width: int) -> str:
    return "horizontal" if height > width else "vertical"


def compose_parts(scene: Image.Image, mask: Image.Image, words: Sequence[str], font_path: Optional[str] = None):
    """-> (glyph, scene, mask as uint8 [h, w, 3] arrays, horizontal, meta): what `compose` stacks.  The rendering of the glyph
    image is the only step that needs the host (font rasteriser); the stacking itself can run on the device
    (ops.compose_canvas / tfx_compose_canvas)."""
    scene, mask = scene.convert("RGB"), mask.convert("RGB")
    if len(words) > 1:
        glyph = render_multiline(scene, mask, words, font_path)
        direction = choose_concat_direction(scene.size[1], scene.size[0])
        meta = dict(mode="multiline", direction=direction)
    else:
        glyph, strip_h = render_single_line(scene, words, font_path)
        direction, meta = "vertical", dict(mode="singleline", direction="vertical", strip=strip_h, orig_h=scene.size[1])
    return np.array(glyph.convert("RGB")), np.array(scene), np.array(mask), direction == "horizontal", meta


def compose(scene: Image.Image, mask: Image.Image, words: Sequence[str], font_path: Optional[str] = None):
    """-> (combined image, combined mask, meta).  Glyph image goes first (top / left); its mask is black."""
    g, s, m, horizontal, meta = compose_parts(scene, mask, words, font_path)
    stack = np.hstack if horizontal else np.vstack
    return Image.fromarray(stack((g, s))), Image.fromarray(stack((np.zeros_like(g), m))), meta


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
