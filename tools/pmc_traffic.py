"""HBM traffic per launch from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only):
    python tools/pmc_traffic.py <fetch.db> <write.db> [workload perceptual(0|1) steps] > profiles/<round>_pmc_traffic_<workload>.json
bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (FETCH_SIZE doubled per /opt/skills/guides/MI355X_MICROARCH.md: on gfx950 the
counter expression tallies 128-byte read requests of wide coalesced streams at 64 bytes; WRITE_SIZE used as reported)."""
import collections
import json
import re
import sqlite3
import sys


def hx_tile(name):
    """(BN, waves) of a k_conv_hx<T, NPL, TH, TW, BN, WM, WN, D, EP, IO> instance from any spelling rocprofv3 produces: fully mangled
    (..Li8ELi16ELi64ELi2ELi2E..), or half-demangled, where leading integers are eaten ("ELi1, ELi64E, 2, 2, 3, 0, 0" = TH 8, TW 16, BN 64, WM 2, WN 2;
    "ELi16ELi64ELi4EL, int, E, 3, 0, 0" = BN 64, WM 4, WN 1 spelled "L, int, E").  BN is the first 32 / 64 / 128 (TH, TW are 4 / 8 / 16), WM and WN the two
    tokens behind it."""
    m = re.search(r"Li(32|64|128)E", name)
    if not m:
        return None, None
    toks = re.findall(r"Li(\d+)E|(L, int, E)|, (\d+)", name[m.end():])
    vals = [int(a or c) if (a or c) else 1 for a, b, c in toks]
    if len(vals) < 2:
        return int(m.group(1)), None
    return int(m.group(1)), vals[0] * vals[1]


def short(name):
    if "k_conv_hx" in name:      # hipcc leaves these template kernels mangled in the trace and rocprofv3 half-demangles the bf16 instances ("<bool _Accum, int, EL, ...>")
        bn, waves = hx_tile(name)
        if bn is None:
            return "k_conv_hx<?>"
        return f"k_conv_hx<{bn}, 8 waves>" if (bn == 128 and waves == 8) else f"k_conv_hx<{bn}>"
    if "k_wgrad_hx" in name:
        return "k_wgrad_hx"
    n = re.sub(r"\(anonymous namespace\)::", "", name)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"k_map<(\w+)>", n)
    if m:
        return "k_map<" + m.group(1) + ">"
    n = n.split("(")[0]
    n = re.sub(r"^(k_conv_thin_out|k_conv_thin_in|k_wgrad_thin|k_conv_wgrad_small|k_conv_wgrad_tile|k_conv_narrow)<.*>$", r"\1", n)
    return n


def collect(db, counter):
    c = sqlite3.connect(db)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for name, val in c.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        a = agg[short(name)]
        a[0] += 1
        a[1] += val
    return agg


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {"_doc": __doc__.strip().replace("\n", " "), "workload": sys.argv[3] if len(sys.argv) > 3 else None,
       "perceptual": (sys.argv[4] == "1") if len(sys.argv) > 4 else None, "kernels": {}}
for k in sorted(fetch):
    n, f = fetch[k]
    w = write.get(k, [n, 0.0])[1]
    out["kernels"][k] = {"launches": n, "fetch_kb_per_launch": f / n, "write_kb_per_launch": w / n, "hbm_bytes_per_launch": (2 * f + w) * 1024 / n}
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 2      # training steps inside the profiled command (gpu_pmc.sh: 1 warm-up + 1 timed)
out["steps_profiled"] = steps
out["total_hbm_bytes_per_step"] = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in out["kernels"].values()) / steps
print(json.dumps(out, indent=1))
