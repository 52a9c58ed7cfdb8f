"""Loaders and dumpers around the hot path: the same public names, arguments, file formats and error behaviour
as the reference's helpers module (reference helpers.py:20-181), re-implemented for an HBM-resident pipeline.

What differs underneath (results are identical):
  * render_strings runs ONE batched forward for all strings instead of a B=1 forward per string
    (reference helpers.py:50-68 streams the 491 MB fc_output weight once per string);
  * load_string_dataset can also hand back the targets as uint8 (the BMPs are 8-bit: 2.9 GB for 150 k sheets
    instead of 11.5 GB of float32), which is what the training loop keeps in HBM.
"""
import os

import numpy as np
import torch
import torch.utils.data as data
from PIL import Image

MODEL_FILENAME = "font_renderer.pth"          # reference helpers.py:18


def binary_array_to_image(binary_array, output_path=None):
    """[H,W] floats in 0..1 (0 black, 1 white) -> 8-bit PIL image, optionally saved as BMP.
    Quantisation truncates, as the reference does (helpers.py:33): 0.999 -> 254."""
    pixels = (np.asarray(binary_array) * 255).astype(np.uint8)
    image = Image.fromarray(pixels)
    if output_path:
        os.makedirs(os.path.dirname(output_path) or ".", exist_ok=True)
        image.save(output_path, "BMP")
    return image


def encode_for_model(strings, max_length, warn=True):
    """ord() codes, cut to max_length (with the reference's warning, helpers.py:52-54), zero padded (:57-59)."""
    rows = np.zeros((len(strings), max_length), dtype=np.int64)
    for i, s in enumerate(strings):
        if len(s) > max_length:
            s = s[:max_length]
            if warn:
                print(f"Warning: String truncated to {max_length} characters: {s}")
        rows[i, :len(s)] = [ord(c) for c in s]
    return rows


def render_strings(model, strings, output_dir, sheet_height, sheet_width, device):
    """Render strings to {output_dir}/string_{i}.bmp (8-bit BMP), reference helpers.py:46-74."""
    os.makedirs(output_dir, exist_ok=True)
    codes = torch.from_numpy(encode_for_model(strings, model.max_length))
    was_training = model.training
    model.eval()
    with torch.no_grad():
        sheets = model(codes.to(device))                       # [N, H, W], one batched launch sequence
    if was_training:
        model.train()
    sheets = sheets.detach().float().cpu().numpy()
    for idx in range(len(strings)):
        binary_array_to_image(sheets[idx], output_path=f"{output_dir}/string_{idx}.bmp")
    print(f"Saved {len(strings)} rendered strings to {output_dir}/")


def save_model(model, filename=MODEL_FILENAME):
    """state_dict (12 float32 tensors, reference key names) -> torch file; interchangeable with the reference's."""
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, filename)
    print(f"Model saved to {filename}")


def load_model(model_class, max_length, filename=MODEL_FILENAME, device=None):
    """Instantiate model_class(max_length=...), load weights, return it in eval mode (reference helpers.py:81-105)."""
    model = model_class(max_length=max_length)
    if device is None:
        device = torch.device("cpu")
    state = torch.load(filename, map_location="cpu", weights_only=True)
    model.load_state_dict(state)
    model = model.to(device)
    model.eval()
    print(f"Model loaded from {filename}")
    return model


def image_to_binary_array(image_path):
    """BMP -> float32 [H,W] in 0..1 via PIL 'L' conversion and /255.0 (reference helpers.py:107-123)."""
    img = Image.open(image_path).convert("L")
    return np.array(img, dtype=np.float32) / 255.0


def image_to_u8_array(image_path):
    return np.array(Image.open(image_path).convert("L"), dtype=np.uint8)


def _read_strings(data_dir, num_samples):
    strings_path = os.path.join(data_dir, "data.txt")
    with open(strings_path, "r") as f:
        strings = f.read().splitlines()
    if len(strings) < num_samples:
        raise ValueError(f"Not enough strings in {strings_path}. Expected {num_samples}, got {len(strings)}")
    return strings


def _load_arrays(data_dir, num_samples, sheet_height, sheet_width, as_uint8):
    strings = _read_strings(data_dir, num_samples)
    targets = np.zeros((num_samples, sheet_height, sheet_width), dtype=np.uint8 if as_uint8 else np.float32)
    codes = []
    for i in range(num_samples):
        image_path = os.path.join(data_dir, f"{i + 1}.bmp")               # 1-based file names, generate_font.ts:210
        if not os.path.exists(image_path):
            raise FileNotFoundError(f"Image file not found: {image_path}")
        targets[i] = image_to_u8_array(image_path) if as_uint8 else image_to_binary_array(image_path)
        codes.append([ord(c) for c in strings[i]])
    max_len = max(len(c) for c in codes)
    inputs = np.zeros((num_samples, max_len), dtype=np.int64)
    for i, c in enumerate(codes):
        inputs[i, :len(c)] = c
    return inputs, targets


def load_string_dataset(data_dir="train_input", num_samples=50000, sheet_height=80, sheet_width=240):
    """TensorDataset(int64 [N, max_len] codes, float32 [N,H,W] targets), reference helpers.py:125-181."""
    print(f"Loading {num_samples} samples from {data_dir}...")
    inputs, targets = _load_arrays(data_dir, num_samples, sheet_height, sheet_width, as_uint8=False)
    print(f"Dataset loading complete: {num_samples} samples with dimensions {sheet_height}x{sheet_width}")
    return data.TensorDataset(torch.from_numpy(inputs), torch.from_numpy(targets))


def load_string_dataset_u8(data_dir="train_input", num_samples=50000, sheet_height=80, sheet_width=240):
    """Same files, targets kept as the 8-bit pixels they are: TensorDataset(int64 codes, uint8 [N,H,W])."""
    print(f"Loading {num_samples} samples from {data_dir}...")
    inputs, targets = _load_arrays(data_dir, num_samples, sheet_height, sheet_width, as_uint8=True)
    print(f"Dataset loading complete: {num_samples} samples with dimensions {sheet_height}x{sheet_width}")
    return data.TensorDataset(torch.from_numpy(inputs), torch.from_numpy(targets))


def targets_as_uint8(targets):
    """float32 targets that are exactly k/255 (what image_to_binary_array produces) -> uint8 k; else None."""
    if targets.dtype == torch.uint8:
        return targets
    q = torch.round(targets * 255.0)
    if bool(((q / 255.0) == targets).all()) and float(q.min()) >= 0 and float(q.max()) <= 255:
        return q.to(torch.uint8)
    return None
