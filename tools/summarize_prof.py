#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/prof_*) into the small summaries
committed under profiles/.  Usage: tools/summarize_prof.py [--between] <tag> <stats_dir> [fetch_dir write_dir mfma_dir]
--between: only the dispatches between bench.py's two hf_profile_marker_kernel launches (its timed region)."""
import collections
import csv
import os
import re
import sys


def short(name):
    m = re.search(r"conv_mfma_hILi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELb(\d)ELb(\d)ELi(\d+)ELb(\d)ELb(\d)E", name)
    if m:  # anonymous-namespace templates come out mangled
        nt, ct, pg, wc, wp, mod, up, tw, pre, fuse = (int(v) for v in m.groups())
        return (f"conv_mfma_h<NTERMS={nt},{ct},{pg},{wc},{wp}{',tw%d' % tw if tw > 32 else ''}{',up' if up else ''}"
                f"{',pre' if pre else ''}{',fuse' if fuse else ''}>")
    if "blur4x4_split8" in name:
        return "blur4x4_split8"
    ms = re.match(r"_Z\d+small_up_blurILb(\d)E", name)
    if ms:
        return "small_up_blur<split>" if ms.group(1) == "1" else "small_up_blur<fp32>"
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
    if "split_weights" in name:
        return "split_weights"
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:60]


MARKER = "hf_profile_marker_kernel"


def marker_window(rows, grid_key):
    """[end of the id-1 marker, start of the id-2 marker] in rocprofv3 timestamps (bench.py launches
    hf_profile_marker_kernel with id + 1 workgroups of 64 around its timed region), or None without markers."""
    t0 = t1 = None
    for r in rows:
        if MARKER in r["Kernel_Name"]:
            wg = int(r[grid_key]) // 64
            if wg == 2:
                t0 = int(r["End_Timestamp"])
            elif wg == 3 and t0 is not None:
                t1 = int(r["Start_Timestamp"])
    return None if t0 is None or t1 is None else (t0, t1)


def region_stats(stats_dir):
    """Per-kernel table of the dispatches inside the marker window of <stats_dir>/bench_kernel_trace.csv."""
    rows = list(csv.DictReader(open(os.path.join(stats_dir, "bench_kernel_trace.csv"))))
    win = marker_window(rows, "Grid_Size_X")
    if win is None:
        sys.exit("--between: no hf_profile_marker_kernel pair in the kernel trace")
    acc = collections.defaultdict(list)
    for r in rows:
        if win[0] <= int(r["Start_Timestamp"]) <= win[1] and MARKER not in r["Kernel_Name"]:
            acc[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    total = sum(sum(v) for v in acc.values())
    table = [(k, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3, 100.0 * sum(v) / total) for k, v in acc.items()]
    return sorted(table, key=lambda t: -t[5]), total / 1e6, (win[1] - win[0]) / 1e6, sum(len(v) for v in acc.values())


def main():
    between = "--between" in sys.argv
    if between:
        sys.argv.remove("--between")
    tag, stats = sys.argv[1], sys.argv[2]
    cmd = os.environ.get("PROFILE_CMD", "python bench.py --steps 3 --warmup 1")
    out = [f"# rocprofv3 summary {tag}", "", f"## --kernel-trace --stats ({cmd})", ""]
    if between:
        table, kernel_ms, wall_ms, n = region_stats(stats)
        out += [f"Dispatches between the two `hf_profile_marker_kernel` launches of bench.py's timed region only (warm-up and plan-time "
                f"kernels excluded): {n} dispatches, {kernel_ms:.1f} ms of kernel time in a {wall_ms:.1f} ms window.", "",
                "| kernel | calls | avg us | min us | max us | % of GPU time |", "|---|---|---|---|---|---|"]
        for k, calls, avg, mn, mx, pct in table:
            if pct >= 0.05:
                out.append(f"| {k} | {calls} | {avg:.1f} | {mn:.1f} | {mx:.1f} | {pct:.2f} |")
    else:
        out += ["| kernel | calls | avg us | min us | max us | % of GPU time |", "|---|---|---|---|---|---|"]
        for r in csv.DictReader(open(os.path.join(stats, "bench_kernel_stats.csv"))):
            if float(r["Percentage"]) < 0.05:
                continue
            out.append(f"| {short(r['Name'])} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | "
                       f"{float(r['MinNs']) / 1e3:.1f} | {float(r['MaxNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
    if len(sys.argv) > 3:
        def agg(d, counter):
            acc = collections.defaultdict(lambda: [0.0, 0])
            rows = list(csv.DictReader(open(os.path.join(d, "bench_counter_collection.csv"))))
            win = marker_window(rows, "Grid_Size") if between else None
            for r in rows:
                if win is not None and not (win[0] <= int(r["Start_Timestamp"]) <= win[1]):
                    continue
                if r["Counter_Name"] == counter:
                    a = acc[short(r["Kernel_Name"])]
                    a[0] += float(r["Counter_Value"])
                    a[1] += 1
            return acc
        f, w = agg(sys.argv[3], "FETCH_SIZE"), agg(sys.argv[4], "WRITE_SIZE")
        busy, gui = agg(sys.argv[5], "SQ_VALU_MFMA_BUSY_CYCLES"), agg(sys.argv[5], "GRBM_GUI_ACTIVE")
        out += ["", "## PMC passes (separate runs of the same command; averages per launch" + (", timed region only" if between else "") + ")", "",
                "FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; FETCH_SIZE is NOT doubled here "
                "(MI355X_MICROARCH.md: it under-counts wide 16 B/lane streaming reads by 2x; uncalibrated for the "
                "dword halo loads of modconv).  MFMA util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs).",
                "", "| kernel | launches | FETCH MB | WRITE MB | MFMA util |", "|---|---|---|---|---|"]
        for k in sorted(f, key=lambda k: -f[k][0]):
            if f[k][0] / f[k][1] < 1000 and k not in busy:
                continue
            util = ""
            if k in busy and gui[k][0] > 0 and busy[k][0] > 0:
                util = f"{busy[k][0] / (gui[k][0] / 8 * 1024):.3f}"
            out.append(f"| {k} | {f[k][1]} | {f[k][0] / f[k][1] / 1024:.1f} | "
                       f"{(w[k][0] / max(1, w[k][1])) / 1024:.1f} | {util} |")
    print("\n".join(out))


if __name__ == "__main__":
    main()
