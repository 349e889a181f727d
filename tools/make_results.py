"""One source of truth for the numbers quoted in README.md / DESIGN.md: the block between
`<!-- results:begin -->` and `<!-- results:end -->` in both files is regenerated from the bench lines stored under
profiles/ (r03_bench.json = `python bench.py --steps 20` on one MI355X; the other files are the same script with the flags
named in the table).      python tools/make_results.py"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def load(name):
    path = os.path.join(P, name)
    if not os.path.exists(path):
        return None
    for line in open(path):
        if line.startswith("{"):
            return json.loads(line)
    return None


def main():
    d = load("r03_bench.json")
    k = d["kernel_ms_per_step"]
    r = d["roofline"]
    cpu = d["cpu_baseline"]
    rows = []
    rows.append("| | detections/s | ms / step |")
    rows.append("|---|---|---|")
    rows.append(f"| **`value`: bit-exact path, 64 x 640x480 frames resident in HBM** (`python bench.py --steps 20`; one batch in flight ahead of the one being collected) | **{d['value']:.0f}** | {d['ms_per_step']:.2f} |")
    hi = d["host_input"]
    rows.append(f"| `host_input`: frames in pageable host memory -> candidates in host memory (`pbd_detect_batch_submit` / `_wait`) | {hi['value']:.0f} ({100 * hi['vs_device_resident']:.1f} % of `value`) | {hi['ms_per_step']:.2f} |")
    rows.append(f"| the same through the synchronous `pbd_detect_batch` | {hi['synchronous']['value']:.0f} | {hi['synchronous']['ms_per_step']:.2f} |")
    fm = d["fast_mode"]
    ag = d["agreement"]
    rows.append(f"| `fast_mode`: `PBD_CONV_MFMA` (bf16 hi/lo split on the matrix cores, responses within 1e-4; {ag['common_with_identical_parts']} of {ag['candidates_exact']} candidates identical to the exact run) | {fm['value']:.0f} | {fm['ms_per_step']:.2f} |")
    f16 = load("r03_bench_f16.json")
    if f16:
        rows.append(f"| `--conv-mode mfma_f16` (BASELINE configs[4]: fp16 operands and responses) | {f16['value']:.0f} | {f16['ms_per_step']:.2f} |")
    rc = load("r03_bench_rccl1.json")
    if rc:
        rows.append(f"| `--force-collective`: the multi-GPU step on one GPU (torch.distributed `nccl`, world 1, one RCCL all_gather of the device-resident payload per step) | {rc['value']:.0f} | {rc['ms_per_step']:.2f} |")
    for name, label in (("r03_bench_hd8.json", "8 x 1920x1080 (BASELINE configs[3] frame size)"), ("r03_bench_hd16.json", "16 x 1920x1080")):
        h = load(name)
        if h:
            extra = f" / MFMA {h['fast_mode']['value']:.0f}" if h.get("fast_mode") else ""
            rows.append(f"| {label}, exact{extra} | {h['value']:.1f}{extra and ''} | {h['ms_per_step']:.2f} |")
    b1 = load("r03_bench_b1.json")
    if b1:
        rows.append(f"| one 640x480 frame per step (`--batch 1`) | {b1['value']:.0f} | {b1['ms_per_step']:.2f} |")
    for name, label in (("r03_bench_b1s4.json", "one 640x480 frame per step over 4 handles (`--batch 1 --streams 4`, `detector.DetectorPool`)"),
                        ("r03_bench_hd1.json", "one 1920x1080 frame per step"),
                        ("r03_bench_hd1s3.json", "one 1920x1080 frame per step over 3 handles (`--streams 3`)"),
                        ("r03_bench_hd8s3.json", "8 x 1920x1080 per step over 3 handles (`--streams 3`)")):
        h = load(name)
        if h:
            rows.append(f"| {label} | {h['value']:.1f} | {h['ms_per_step']:.2f} |")
    o3 = cpu.get("O3") or {}
    rows.append(f"| CPU restatement, {cpu['cores']} OpenMP threads of the GPU box's host (`cpu_baseline`, kind \"port\"): -O2 / -O3 / one thread | {cpu['value']:.2f} / {o3.get('value', float('nan')):.2f} / {cpu['single_thread']['value']:.2f} | - |")
    rows.append("")
    rows.append(f"Per-kernel ms per step (HIP events, `kernel_ms_per_step`): " + ", ".join(f"{n[2:]} {v:.2f}" for n, v in k.items()) + ".")
    rows.append(f"`roofline` (dominant kernel `{r['kernel']}`, {r['avg_launch_ms']:.2f} ms per launch measured inside the timed region): "
                f"{r['achieved']:.1f} TFLOP/s = **{r['frac']:.3f}** of the {r['peak']} TFLOP/s fp32 peak (ceiling 0.50: two separately rounded lane-ops per MAC); "
                f"HBM traffic {r['traffic'] / 1e9:.1f} GB per launch ({r['wasted_traffic']:.2f}x the {r['algorithmic_bytes_per_launch'] / 1e9:.2f} GB algorithmic).")
    for e in d["roofline_all"][1:]:
        if e.get("bound") == "hbm":
            rows.append(f"`{e['kernel']}`: {e['achieved'] / 1e3:.2f} TB/s of plane I/O = {e['frac']:.2f} of HBM, traffic {('%.1fx' % e['wasted_traffic']) if e['wasted_traffic'] else 'n/a'}.")
    block = "\n".join(rows)
    for fn in ("README.md", "DESIGN.md"):
        path = os.path.join(ROOT, fn)
        s = open(path).read()
        if "<!-- results:begin -->" not in s:
            continue
        s = re.sub(r"<!-- results:begin -->.*?<!-- results:end -->", "<!-- results:begin -->\n" + block + "\n<!-- results:end -->", s, flags=re.S)
        open(path, "w").write(s)
    print(block)


if __name__ == "__main__":
    main()
