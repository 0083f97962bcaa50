"""Generate the golden fixtures under tests/golden/.

ORACLE-GENERATED, NOT REFERENCE-GENERATED: the reference cannot run in this
image (import cv2 fails) and ships no vectors of its
own, so these files pin the *oracle* (oracle/wm_oracle.py, NumPy 2.2.6 /
SciPy 1.15.3) against drift, and give the GPU tests fixed inputs/outputs.

    python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import wm_oracle as o  # noqa: E402

PASSWORD = "golden-pw"
NONCE = bytes(range(8))

# name, H, W, tile, alpha, kfrac, k_floor, color
CASES = [
    ("gray_16x16_ref", 16, 16, None, 0.12, 0.6, 8, False),
    ("gray_64x64_ref", 64, 64, None, 0.12, 0.6, 8, False),
    ("gray_64x96_ref", 64, 96, None, 0.15, 0.6, 8, False),     # non-square: the [:L,:L] quirk
    ("gray_16x16_t8", 16, 16, 8, 0.12, 0.6, 8, False),
    ("gray_64x96_t8", 64, 96, 8, 0.15, 0.6, 8, False),
    ("gray_64x64_t8_k3", 64, 64, 8, 0.15, 0.0, 3, False),      # mid-band: only 3 singular values
    ("gray_45x70_t8", 45, 70, 8, 0.12, 0.6, 8, False),         # ragged border
    ("color_32x48_t8", 32, 48, 8, 0.18, 0.6, 8, True),
    ("color_32x32_ref", 32, 32, None, 0.18, 0.6, 8, True),
]


def build(name, H, W, tile, alpha, kfrac, k_floor, color):
    rng = np.random.default_rng(int(hashlib.sha256(name.encode()).hexdigest()[:8], 16))
    cover = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    wm = rng.integers(0, 256, (max(H // 4, 2), max(W // 4, 2), 3), dtype=np.uint8)
    r = o.embed_arrays(cover, wm, PASSWORD, NONCE, alpha, color, kfrac, tile, k_floor)
    ex = o.extract_arrays(r["stego"], r["meta"], PASSWORD, True, tile, k_floor)
    ok, score = o.detect_arrays(r["stego"], r["meta"], 0.6, tile)
    key = o.derive_key(PASSWORD, NONCE)
    idx = o.permutation(H, W, o.rng_from_key(key))
    out = dict(cover=cover, wm=wm, stego=r["stego"], extracted=ex, detect_score=np.float64(score),
               psnr=np.float64(r["psnr"]), ssim=np.float64(r["ssim"]),
               perm_sha256=np.frombuffer(hashlib.sha256(idx.astype(np.int64).tobytes()).digest(), np.uint8),
               alpha=np.float64(alpha), kfrac=np.float64(kfrac), k_floor=np.int32(k_floor),
               tile=np.int32(-1 if tile is None else tile), color=np.bool_(color))
    for k, v in r["meta"].items():
        if k in ("mode", "payload_type"):
            continue
        # singular-vector matrices of the full-frame cases are large and sign-ambiguous:
        # keep singular values + digest, drop U/V (the digest covers their bytes)
        if tile is None and (k.startswith("U") or k.startswith("V")) and H * W > 1024:
            continue
        out["meta_" + k] = np.asarray(v)
    return out


def main():
    for case in CASES:
        d = build(*case)
        np.savez_compressed(os.path.join(HERE, case[0] + ".npz"), **d)
        print(case[0], sum(v.nbytes for v in d.values()), "bytes uncompressed")


if __name__ == "__main__":
    main()
