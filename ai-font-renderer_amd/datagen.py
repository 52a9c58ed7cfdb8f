"""Dataset generator: what the reference's offline generate_font.ts produces (train_input/{1..N}.bmp + data.txt +
dataset_metadata.txt), re-implemented on CPU with PIL/FreeType because bun + node-canvas (cairo) are not part of
this stack.  Text source and file formats are exact restatements; the rasterisation is FreeType's, so pixels are
close to but not bit-identical with cairo's -- irrelevant to the training path, for which targets are inputs.

  python -m ai_font_renderer_amd.datagen --font FiraCode-Retina.ttf [--n 150000] [--out train_input]
"""
import argparse
import os
import struct

import numpy as np

from .synth import lcg_text

FONT_SIZE = 12                  # generate_font.ts:66
SHEET_WIDTH, SHEET_HEIGHT = 240, 80
PADDING = 0
LINE_HEIGHT = FONT_SIZE * 1.2   # generate_font.ts:125


def bmp24_topdown(rgb):
    """uint8 [H,W,3] RGB -> the file generate_font.ts:6-62 writes: 54-byte header, 24 bpp BGR, negative height
    (top-down), rows padded to 4 bytes."""
    h, w, _ = rgb.shape
    row = (w * 3 + 3) // 4 * 4
    body = np.zeros((h, row), dtype=np.uint8)
    body[:, :w * 3] = rgb[:, :, ::-1].reshape(h, w * 3)
    data = body.tobytes()
    return (b"BM" + struct.pack("<IHHI", 54 + len(data), 0, 0, 54)
            + struct.pack("<IiiHHIIiiII", 40, w, -h, 1, 24, 0, len(data), 0, 0, 0, 0) + data)


def wrap_text(measure, text, max_width):
    """Greedy word wrap by measured width (generate_font.ts:75-97)."""
    lines, current = [], ""
    for word in text.split(" "):
        test = f"{current} {word}" if current else word
        if measure(test) > max_width and current:
            lines.append(current)
            current = word
        else:
            current = test
    if current:
        lines.append(current)
    return lines


def render_sheet(text, font):
    """White sheet, black text, baseline of line i at y=(i+1)*14.4 (generate_font.ts:112-130).  uint8 [80,240]."""
    from PIL import Image, ImageDraw
    img = Image.new("L", (SHEET_WIDTH, SHEET_HEIGHT), 255)
    draw = ImageDraw.Draw(img)
    for i, line in enumerate(wrap_text(font.getlength, text, SHEET_WIDTH - 2 * PADDING)):
        draw.text((PADDING, PADDING + (i + 1) * LINE_HEIGHT), line, fill=0, font=font, anchor="ls")
    return np.asarray(img, dtype=np.uint8)


def render_glyphs(font_paths, size, codes=range(32, 127)):
    """Per-glyph targets of the BASELINE glyph configs: for every font and every printable ASCII code one size x size
    bitmap, white background, black glyph (the generate_font.ts:112-142 idiom -- fillStyle white, fillText black -- at
    glyph scale): the glyph is drawn with its advance box centred horizontally and the font's ascent/descent box centred
    vertically, at a pixel size of 0.8 * size.  uint8 [n_fonts, n_codes, size, size]."""
    from PIL import Image, ImageDraw, ImageFont
    out = np.zeros((len(font_paths), len(codes), size, size), dtype=np.uint8)
    for fi, path in enumerate(font_paths):
        font = ImageFont.truetype(path, int(round(size * 0.8)))
        ascent, descent = font.getmetrics()
        y0 = (size - (ascent + descent)) / 2.0 + ascent
        for ci, code in enumerate(codes):
            img = Image.new("L", (size, size), 255)
            ch = chr(code)
            x0 = (size - font.getlength(ch)) / 2.0
            ImageDraw.Draw(img).text((x0, y0), ch, fill=0, font=font, anchor="ls")
            out[fi, ci] = np.asarray(img, dtype=np.uint8)
    return out


def generate(out_dir, n, font_path=None, first_seed=42):
    """font_path=None falls back to Pillow's built-in scalable font (tests; the real data set uses Fira Code)."""
    from PIL import ImageFont
    font = ImageFont.truetype(font_path, FONT_SIZE) if font_path else ImageFont.load_default(FONT_SIZE)
    os.makedirs(out_dir, exist_ok=True)
    texts = [lcg_text(first_seed + i, 10, 100) for i in range(n)]       # generate_font.ts:203-206
    with open(os.path.join(out_dir, "data.txt"), "w") as f:
        f.write("\n".join(texts))                                        # no trailing newline, generate_font.ts:216
    for i, t in enumerate(texts):
        g = render_sheet(t, font)
        with open(os.path.join(out_dir, f"{i + 1}.bmp"), "wb") as f:     # 1-based names, generate_font.ts:210
            f.write(bmp24_topdown(np.stack([g, g, g], axis=-1)))
    with open(os.path.join(out_dir, "dataset_metadata.txt"), "w") as f:
        f.write(f"AI Font Renderer Dataset - Fira Code\n==============================\n\nFont: {font_path}\n"
                f"Font size: {FONT_SIZE}\nSheet dimensions: {SHEET_WIDTH}x{SHEET_HEIGHT}\nPadding: {PADDING}px\n\n"
                "Format: Images are numbered sequentially (1.bmp, 2.bmp, etc.)\n"
                "Text labels are stored line by line in data.txt (line 1 corresponds to 1.bmp)\n")
    return texts


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--font", required=True)
    ap.add_argument("--n", type=int, default=150000)
    ap.add_argument("--out", default="train_input")
    a = ap.parse_args()
    generate(a.out, a.n, a.font)
    print(f"Dataset generation complete. Check the {a.out}/ directory.")
