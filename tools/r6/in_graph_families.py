"""rocprofv3 --kernel-trace --stats summary of the bench command (tools/summarize_rocprof.py output: kernel,calls,total_ms,...) ->
per-family kernel time INSIDE the replayed graph, per forward:  python tools/r6/in_graph_families.py <stats.csv> <forwards> <out.json>

The bench's own `families` come from an eager pass with a HIP event pair around every launch (short launches read 40 - 60 % slow there);
these are the durations of the same kernels as the graph replay runs them.  Kernel -> family by name; the generic implicit-GEMM kernels
(igemm_kernel / igemm_glds_kernel: stride-2 and odd-shaped convs AND some 1x1 GEMMs) and the split-K reduce passes (conv and GEMM
launches alike) cannot be split by name and are reported as their own rows."""
import csv
import json
import sys

RULES = [("igemm_halo", "conv3x3_igemm"), ("conv3x3_wstream", "conv3x3_igemm"), ("gemm_glds_kernel", "gemm1x1_igemm"),
         ("igemm_glds_kernel", "igemm_generic(conv|1x1)"), ("igemm_kernel", "igemm_generic(conv|1x1)"), ("splitk_reduce", "splitk_reduce(conv|1x1)"),
         ("attn", "attention"), ("tchain_head", "chain_head"), ("tchain_tail", "chain_tail"), ("tchain_csce", "chain_csce"), ("tchain_mlp", "chain_mlp"),
         ("gn_finalize", "groupnorm_finalize"), ("gn_stats", "groupnorm_stats"), ("gn_apply", "groupnorm"), ("ln_rows", "layernorm"),
         ("softmax_rows", "softmax"), ("dwconv3x3", "dwconv3x3"), ("linear_f32", "linear_f32"), ("prefetch_lines", "weight_prefetch_branch")]


def family(name):
    for key, fam in RULES:
        if key in name:
            return fam
    return "elementwise+torch"


def main():
    src, forwards, dst = sys.argv[1], float(sys.argv[2]), sys.argv[3]
    fams = {}
    for r in csv.DictReader(open(src)):
        f = fams.setdefault(family(r["kernel"]), {"launches": 0.0, "ms": 0.0})
        f["launches"] += float(r["calls"]) / forwards
        f["ms"] += float(r["total_ms"]) / forwards
    out = {"source": f"{src}: rocprofv3 --kernel-trace --stats of `python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-fp16 "
                     f"--no-other-configs`, totals / {forwards:g} forwards (1 eager warm-up pass + the graph replays)",
           "families_in_graph": {k: {"launches": round(v["launches"], 1), "ms": round(v["ms"], 3)} for k, v in sorted(fams.items(), key=lambda kv: -kv[1]["ms"])},
           "sum_ms": round(sum(v["ms"] for v in fams.values()), 2)}
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out["families_in_graph"]), out["sum_ms"])


if __name__ == "__main__":
    main()
