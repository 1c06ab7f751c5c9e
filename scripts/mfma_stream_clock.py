"""The shader clock and socket power the chip sustains under a register-only MFMA stream (scripts/probes/mfma_peak.hip:
`v_mfma_f32_32x32x16_bf16` back to back on every SIMD, optionally with exp / fma between the MFMAs), sampled with bench.py's
ClockPowerSampler while the stream runs -- the "what the matrix pipe alone gets" row next to the flash kernel's own
`roofline.sclk_mhz_mean` (review round 5, item 2).  Compiles the probe on the box (hipcc, ~5 s).

    python scripts/mfma_stream_clock.py [--seconds 2] > profiles/r06_x_mfma_stream_clock.jsonl
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

NAMES = {0: 'mfma only', 1: 'mfma + 2 exp + 2 fma per mfma', 2: 'mfma + 4 exp + 8 fma per mfma',
         3: 'mfma + 1 KB of LDS reads (one ds_read_b128 per wave) per mfma',
         4: 'mfma + 1 KB of LDS reads + 2 exp + 6 fma per mfma (the instruction mix of an attention tile, registers and LDS only)'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=2.0)
    a = ap.parse_args()
    src = os.path.join(ROOT, 'scripts', 'probes', 'mfma_peak.hip')
    exe = '/tmp/mfma_peak.bin'
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O2', '-w', src, '-o', exe], check=True)
    for mode in (0, 1, 2, 3, 4):
        smi = bench.ClockPowerSampler(0, period_s=0.01).start()
        r = subprocess.run([exe, 'sustain', str(mode), str(a.seconds)], capture_output=True, text=True)
        clocks = smi.stop()
        row = json.loads(r.stdout.strip().splitlines()[-1])
        row.update(stream=NAMES[mode], mfma_frac_of_2500=round(row['tflops'] / 2500.0, 4), **clocks)
        # the sampling window opens before the process has a kernel on the GPU: keep the samples taken under load
        busy = [(c, w) for c, w in smi.samples if c and clocks['sclk_mhz_max'] and c >= 0.8 * clocks['sclk_mhz_max']]
        if busy:
            row['sclk_mhz_under_load'] = round(sum(c for c, _ in busy) / len(busy), 1)
            pw = [w for _, w in busy if w]
            row['power_w_under_load'] = round(sum(pw) / len(pw), 1) if pw else None
            row['mfma_frac_at_sustained_clock'] = round(row['tflops'] / (2500.0 * row['sclk_mhz_under_load'] / 2400.0), 4)
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
