#!/usr/bin/env python3
"""profiles/pmc_traffic.json from whole-bench rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE and
optionally SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE), averaged per launch and keyed by the
kernel labels bench.py uses.  usage: make_pmc_traffic.py <fetch_dir> <write_dir> [<mfma_dir>]"""
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


def agg(d, counter):
    acc = collections.defaultdict(lambda: [0.0, set()])
    for r in csv.DictReader(open(os.path.join(d, "bench_counter_collection.csv"))):
        if r["Counter_Name"] == counter:
            a = acc[label(r["Kernel_Name"])]
            a[0] += float(r["Counter_Value"])
            a[1].add(r["Dispatch_Id"])
    return acc


def main():
    f, w = agg(sys.argv[1], "FETCH_SIZE"), agg(sys.argv[2], "WRITE_SIZE")
    busy = gui = None
    if len(sys.argv) > 3:
        busy, gui = agg(sys.argv[3], "SQ_VALU_MFMA_BUSY_CYCLES"), agg(sys.argv[3], "GRBM_GUI_ACTIVE")
    out = {"tag": os.environ.get("PROFILE_TAG", "untagged"), "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES+GRBM_GUI_ACTIVE (separate passes), "
                     "python bench.py --steps 1 --warmup 1 --no-exact-f32; KiB*1024; FETCH_SIZE doubled for the kernels that read "
                     "16 B / lane (fetch_doubled: LDS-DMA / wide vector loads), as MI355X_MICROARCH.md's HBM section prescribes for "
                     "gfx950; left as counted for the dword halo loads of modconv (uncalibrated); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / "
                     "(GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs), i.e. relative to the clock the kernel actually ran at",
           "kernels": {}}
    for k in f:
        n = max(1, len(f[k][1]))
        if f[k][0] / n < 1000 and not k.startswith("conv_mfma"):
            continue
        # MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide (16 B / lane)
        # coalesced streaming read, `global_load` and `buffer_load ... lds` alike: the kernels whose reads are 16-byte
        # LDS-DMA units (pre-split activations + weights) or 16-byte vector loads get their FETCH doubled
        wide = (",pre" in k and (k.startswith("conv_mfma_h") or k.startswith("conv_enc_h"))) or k in ("blur4x4_split8", "torgb_kernel", "blur4x4_noise_bias_act") or k.startswith("conv_rows_h")
        raw_fetch = f[k][0] / n * 1024
        e = {"fetch_bytes_per_launch": raw_fetch * (2.0 if wide else 1.0), "fetch_size_counter_bytes": raw_fetch,
             "fetch_doubled": bool(wide), "write_bytes_per_launch": w[k][0] / max(1, len(w[k][1])) * 1024, "launches": n}
        if busy and gui and gui[k][0] > 0:
            e["mfma_busy"] = round(busy[k][0] / (gui[k][0] / 8 * 1024), 4)
        out["kernels"][k] = e
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
