#!/usr/bin/env python3
"""Copy the summaries of a tools/prof_round.sh run into profiles/ and refresh
profiles/traffic.json.  Usage: tools/save_profiles.py <gpurun_out dir> <name prefix, e.g. r02a>"""
import json, os, re, shutil, sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
tpath = os.path.join(prof, "traffic.json")
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
for w in ("c1", "c2", "c3", "c4"):
    ks = os.path.join(src, "%s_kernel_stats.csv" % w)
    pm = os.path.join(src, "%s_pmc.txt" % w)
    bj = os.path.join(src, "bench_%s.json" % w)
    if not (os.path.exists(ks) and os.path.exists(pm) and os.path.exists(bj)):
        continue
    shutil.copy(ks, os.path.join(prof, "%s_%s_kernel_stats.csv" % (tag, w)))
    shutil.copy(pm, os.path.join(prof, "%s_%s_pmc.txt" % (tag, w)))
    shutil.copy(bj, os.path.join(prof, "%s_bench_%s.json" % (tag, w)))
    bench = json.load(open(bj))
    kern = bench["roofline"].get("kernel", "")
    vals = {}
    for line in open(pm):
        m = re.match(r"(.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+per-launch\s+([0-9.e+]+)", line)
        if m and (not kern or kern.split("<")[0] in m.group(1)):
            vals[m.group(2)] = float(m.group(3))
    if len(vals) != 2:
        continue
    fetch = vals["FETCH_SIZE"] * 1024 * 2
    write = vals["WRITE_SIZE"] * 1024
    traffic[w] = {
        "records": bench["config"]["records_per_gpu_per_step"],
        "kernel": kern,
        "FETCH_SIZE_KB_per_launch": vals["FETCH_SIZE"],
        "WRITE_SIZE_KB_per_launch": vals["WRITE_SIZE"],
        "fetch_bytes_corrected": fetch,
        "write_bytes": write,
        "bytes_per_launch": fetch + write,
        "algorithmic_bytes_per_launch": bench["roofline"].get("algorithmic_bytes_per_launch"),
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of bench.py --quick on this "
                  "workload, mean over launches; FETCH_SIZE doubled (gfx950 tallies 128-byte read requests at 64 B), "
                  "WRITE_SIZE as read; both count requests of the L2 to the fabric, Infinity Cache hits included; "
                  "the doubling is calibrated for wide streaming reads, so with list-table gathers the read half is "
                  "an upper estimate",
        "source": "profiles/%s_%s_pmc.txt" % (tag, w),
    }
json.dump(traffic, open(tpath, "w"), indent=1)
print(json.dumps({k: (v["bytes_per_launch"], v.get("algorithmic_bytes_per_launch")) for k, v in traffic.items()}, indent=1))
