"""HBM traffic of one decode token step from the rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, separate runs, KB per dispatch):
sum over the kernels of the captured step.  gfx950: FETCH_SIZE counts wide streaming reads at half their bytes (MI355X_MICROARCH.md §HBM),
so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores."""
import json, re, sys
def load(path):
    out = {}
    for line in open(path):
        m = re.match(r"^(.{60})\s+(\d+)\s+(\d+)\s+\w+=([0-9.e+]+)", line)
        if m:
            out[(m.group(1).strip(), int(m.group(2)))] = (int(m.group(3)), float(m.group(4)))
    return out
fetch, write = load(sys.argv[1]), load(sys.argv[2])
layers = int(sys.argv[3]) if len(sys.argv) > 3 else 28
max_new = int(sys.argv[4]) if len(sys.argv) > 4 else 30
per_layer = [k for k in fetch if fetch[k][0] % layers == 0 and fetch[k][0] // layers >= 20 and ("skinny" in k[0] or "decode_attn" in k[0] or "add_rmsnorm" in k[0] or "rmsnorm_ss" in k[0])]
per_step = [k for k in fetch if ("greedy" in k[0] or (("skinny_xs" in k[0]) and k not in per_layer))]
rows = []
tot_f = tot_w = 0.0
for k in per_layer + per_step:
    mult = layers if k in per_layer else 1
    f = fetch[k][1] * 1024 * 2 * mult; w = write.get(k, (0, 0.0))[1] * 1024 * mult
    rows.append({"kernel": k[0], "grid_threads": k[1], "launches_per_step": mult, "fetch_bytes_x2": f, "write_bytes": w})
    tot_f += f; tot_w += w
print(json.dumps({"max_new": max_new, "batch": 32, "mode": "native", "hbm_bytes_per_token_step": tot_f + tot_w, "fetch_bytes_x2": tot_f, "write_bytes": tot_w, "kernels": rows,
                  "note": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --slots 1 --pipeline off --steps 1 --warmup 1 --no-cpu-baseline --no-extras --max-new %d` "
                          "(the per-dispatch averages run over the token steps of that run: at --max-new 150 the contexts are the bench's own, 264..413); FETCH_SIZE doubled (gfx950 correction)" % max_new,
                  "source": [sys.argv[1], sys.argv[2]]}, indent=1))
