"""Per-kernel HBM traffic of the batch-1 decode step from two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; separate
runs, --kernel-trace only, as MI355X_MICROARCH.md prescribes) -> the JSON bench.py reads for `roofline.traffic`.
    tools/pmc_to_json.py <fetch_dir> <write_dir> <out.json>
FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE counts half the bytes of wide coalesced streaming reads
(128-byte requests tallied as 64): reads are doubled, writes taken as they are (guide, section HBM)."""
import collections, csv, glob, json, sys

import re
# fused_gemv_kernel<weights, activations, M, rows per wave, prologue, epilogue, chunks>: classes by (prologue, epilogue)
CLASS = [(re.compile(r"attn_oproj_"), "attn"), (re.compile(r"finalize_kernel"), "argmax"),
         (re.compile(r"fused_gemv_kernel<pgk::bf16, float, 1, \d+, 0, 0,"), "norm_qkv"), (re.compile(r"fused_gemv_kernel<pgk::bf16, float, 1, \d+, 3, 2,"), "gateup"),
         (re.compile(r"fused_gemv_kernel<pgk::bf16, float, 1, \d+, 1, 1,"), "down"), (re.compile(r"fused_gemv_kernel<pgk::bf16, float, 1, \d+, 0, 3,"), "lmhead")]


def collect(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two runs, --kernel-trace only) of "
                 "`python3 bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-extras --long-prefill 0`; reads x2 per MI355X_MICROARCH.md",
       "kernels": {}, "by_name": {}}
for name in sorted(set(fetch) | set(write)):
    fr = fetch.get(name, [0.0]); wr = write.get(name, [0.0])
    rd = 2.0 * 1024.0 * sum(fr) / len(fr); wb = 1024.0 * sum(wr) / len(wr)
    rec = {"launches": len(fr), "FETCH_SIZE_KB_avg": sum(fr) / len(fr), "WRITE_SIZE_KB_avg": sum(wr) / len(wr),
           "hbm_read_bytes_corrected": rd, "hbm_write_bytes": wb, "hbm_bytes_per_launch": rd + wb}
    out["by_name"][name] = rec
    for pat, cls in CLASS:
        if pat.search(name) and (len(fr) >= 20 or (cls in ("lmhead", "argmax") and len(fr) >= 8)):
            out["kernels"][cls] = dict(rec, kernel=name)
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
