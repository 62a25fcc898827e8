#!/usr/bin/env python
"""Static check of the LDS row pitches the split kernels use: a ds_read_b128 is served 16 lanes at a time; the 16 lanes of a
service group read 16 bytes each from 16 consecutive rows (pixels / GEMM rows) at the same in-row offset, so they are conflict
free iff their 16-byte slots fall on 16 distinct positions of the 256-byte (64 banks x 4 bytes) bank window.  No GPU needed.
    python tools/lds_bank_check.py"""
PITCHES = {
    "conv3x3_split bf16x3 pixel / weight row (ROWB)": 112,
    "conv3x3_split f16x3 pixel (ROWB_H, r4)": 80,
    "gemm_split bf16x6 row (GPITCH)": 208,
    "gemm_split f16x3 A row (GPITCH_H)": 144,
    "unpadded f16x3 pixel (64 B: what the padding avoids)": 64,
    "unpadded 128 B row": 128,
}
for name, pitch in PITCHES.items():
    fewest = 16
    for off in range(0, pitch - 15, 16):          # every 16-byte piece of a row
        for r0 in (0, 5, 16):                     # service groups starting at different rows
            slots = [((r0 + i) * pitch + off) % 256 // 16 for i in range(16)]
            fewest = min(fewest, len(set(slots)))
    print(f"{pitch:4d} B  {name:55s} {'conflict-free' if fewest == 16 else f'only {fewest} distinct slots of 16: {16 // fewest}-way conflicts'}")
