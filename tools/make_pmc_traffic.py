#!/usr/bin/env python3
"""profiles/pmc_traffic.json from whole-bench rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE and
optionally SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE), averaged per launch and keyed by the
kernel labels bench.py uses.  usage: make_pmc_traffic.py [--between] <fetch_dir> <write_dir> [<mfma_dir> [<kernel_trace_dir>]]
--between: only the dispatches between bench.py's two hf_profile_marker_kernel launches (its timed region)."""
import collections
import csv
import json
import os
import re
import sys


def label(name):
    m = re.search(r"conv_mfma_hILi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELb(\d)ELb(\d)ELi(\d+)ELb(\d)ELb(\d)E", name)
    if m:
        nt, ct, pg, wc, wp, mod, up, tw, pre, fuse = (int(v) for v in m.groups())
        return (f"conv_mfma_h<{ct},{pg},{wc},{wp}{',tw%d' % tw if tw > 32 else ''}{',up' if up else ''}"
                f"{',pre' if pre else ''}{',fuse' if fuse else ''}>")
    if "blur4x4_split8" in name:
        return "blur4x4_split8"
    ms = re.match(r"_Z\d+small_up_blurILb(\d)E", name)
    if ms:
        return "small_up_blur<split>" if ms.group(1) == "1" else "small_up_blur<fp32>"
    if "conv_rows_h" in name:
        return "conv_rows_h<32->32,strip64,pre>"  # bench.py's label (csrc/convrow.hip)
    mg = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name)
    if mg and "conv_mfma_h" not in name and "conv_enc_h" not in name:  # other anonymous-namespace kernels: the bare identifier
        n = int(mg.group(1))
        ident = name[len(mg.group(0)):len(mg.group(0)) + n]
        targs = re.findall(r"(?:Li(\d+)E|Lb(\d)E)", name[len(mg.group(0)) + n:].split("Ev")[0])
        vals = [a or ("true" if b == "1" else "false") for a, b in targs]
        return ident + (f"<{','.join(vals)}>" if vals else "")
    me = re.search(r"conv_enc_hILi(\d)ELi(\d)ELi(\d)ELi(\d)ELb(\d)E(?:Li(\d)ELi(\d)E)?", name)
    if me:
        nt, pg, wpx, stride, pre = (int(v) for v in me.groups()[:5])
        return f"conv_enc_h<{'64x%d' % (32 * pg * wpx)}{',stride2' if stride == 2 else ''}{',pre' if pre else ''}>"
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    m = re.match(r"(conv_mfma\w*)<(.*)>", name)
    if not m:
        return name
    fam, args = m.group(1), [a.strip() for a in m.group(2).split(",")]
    if fam == "conv_mfma_dma":
        return f"conv_mfma_dma<{','.join(args[:4])}>"
    if fam == "conv_mfma_pipe":
        tag = ",up" if args[4] == "true" else (",stride2" if len(args) > 6 and args[6] == "2" else "")
        return f"conv_mfma_pipe<{','.join(args[:4])}{tag}>"
    return f"{fam}<{','.join(args)}>"


BETWEEN = False
MARKER = "hf_profile_marker_kernel"


def marker_window(rows, grid_key):
    t0 = t1 = None
    for r in rows:
        if MARKER in r["Kernel_Name"]:
            wg = int(r[grid_key]) // 64
            if wg == 2:
                t0 = int(r["End_Timestamp"])
            elif wg == 3 and t0 is not None:
                t1 = int(r["Start_Timestamp"])
    return None if t0 is None or t1 is None else (t0, t1)


def trace_avg_us(d):
    """kernel label -> average duration (us) of the dispatches inside the marker window of a --kernel-trace run."""
    rows = list(csv.DictReader(open(os.path.join(d, "bench_kernel_trace.csv"))))
    win = marker_window(rows, "Grid_Size_X") if BETWEEN else None
    acc = collections.defaultdict(list)
    for r in rows:
        if win is None or win[0] <= int(r["Start_Timestamp"]) <= win[1]:
            acc[label(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return {k: sum(v) / len(v) / 1e3 for k, v in acc.items()}


def agg(d, counter):
    acc = collections.defaultdict(lambda: [0.0, set()])
    rows = list(csv.DictReader(open(os.path.join(d, "bench_counter_collection.csv"))))
    win = marker_window(rows, "Grid_Size") if BETWEEN else None
    if BETWEEN and win is None:
        sys.exit(f"--between: no hf_profile_marker_kernel pair in {d}")
    for r in rows:
        if win is not None and not (win[0] <= int(r["Start_Timestamp"]) <= win[1]):
            continue
        if r["Counter_Name"] == counter:
            a = acc[label(r["Kernel_Name"])]
            a[0] += float(r["Counter_Value"])
            a[1].add(r["Dispatch_Id"])
    return acc


def main():
    global BETWEEN
    if "--between" in sys.argv:  # only the dispatches between bench.py's two profile markers (its timed region)
        BETWEEN = True
        sys.argv.remove("--between")
    f, w = agg(sys.argv[1], "FETCH_SIZE"), agg(sys.argv[2], "WRITE_SIZE")
    busy = gui = None
    if len(sys.argv) > 3:
        busy, gui = agg(sys.argv[3], "SQ_VALU_MFMA_BUSY_CYCLES"), agg(sys.argv[3], "GRBM_GUI_ACTIVE")
    avg_us = trace_avg_us(sys.argv[4]) if len(sys.argv) > 4 else {}
    cmd = os.environ.get("PROFILE_CMD", "python bench.py --steps 1 --warmup 1 --no-exact-f32")
    out = {"tag": os.environ.get("PROFILE_TAG", "untagged"), "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES+GRBM_GUI_ACTIVE (separate passes), "
                     + cmd + ("; dispatches between bench.py's profile markers only (timed region)" if BETWEEN else "") + "; KiB*1024; FETCH_SIZE doubled for the kernels that read "
                     "16 B / lane (fetch_doubled: LDS-DMA / wide vector loads), as MI355X_MICROARCH.md's HBM section prescribes for "
                     "gfx950; left as counted for the dword halo loads of modconv (uncalibrated); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / "
                     "(GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs), i.e. relative to the clock the kernel actually ran at",
           "kernels": {}}
    if os.environ.get("PROFILE_SWAP_BATCH"):  # the batched swap pass: bench.py attaches it only to a run of the same pass size / precision
        out["swap_batch"] = int(os.environ["PROFILE_SWAP_BATCH"])
        out["conv_precision"] = os.environ.get("PROFILE_CONV_PRECISION", "f16x3")
    for k in f:
        n = max(1, len(f[k][1]))
        if f[k][0] / n < 1000 and not k.startswith("conv_mfma"):
            continue
        # MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide (16 B / lane)
        # coalesced streaming read, `global_load` and `buffer_load ... lds` alike: the kernels whose reads are 16-byte
        # LDS-DMA units (pre-split activations + weights) or 16-byte vector loads get their FETCH doubled
        wide = (",pre" in k and (k.startswith("conv_mfma_h") or k.startswith("conv_enc_h"))) or k in ("blur4x4_split8", "torgb_kernel", "blur4x4_noise_bias_act") or k.startswith("conv_rows_h") or k.startswith("conv_enc_s2mt_h")
        raw_fetch = f[k][0] / n * 1024
        e = {"fetch_bytes_per_launch": raw_fetch * (2.0 if wide else 1.0), "fetch_size_counter_bytes": raw_fetch,
             "fetch_doubled": bool(wide), "write_bytes_per_launch": w[k][0] / max(1, len(w[k][1])) * 1024, "launches": n}
        if busy and gui and gui[k][0] > 0:
            e["mfma_busy"] = round(busy[k][0] / (gui[k][0] / 8 * 1024), 4)
        if k in avg_us:
            e["avg_us"] = round(avg_us[k], 2)  # of the --kernel-trace run of the same command (no counters)
        out["kernels"][k] = e
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
