"""Rehearsal of the 8-GPU run on ONE GPU: N rank PROCESSES at the full configs[2] size (26 x 100k x 16, FC[512,256,1], batch 4096
per rank, truncated Zipf ids), ps_shard_step over gloo collectives with host staging.  Since round 4 this is a -m gpu test
(tests/test_gpu_rehearse_n8.py, which also checks step 1 against the PS semantics key by key); this tool runs the same rank
processes for any rank / step count and prints the exchange statistics.     python tools/rehearse_n8.py [ranks=8] [steps=12]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

if __name__ == "__main__":
    from test_gpu_rehearse_n8 import run_ranks
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    res = run_ranks(world, steps, snapshot=False)
    bad = [r for r in res if r[1] != "ok"]
    for r in bad: print("rank %d FAILED:\n%s" % (r[0], r[2]))
    if bad: sys.exit(1)
    d0 = res[0][2]["digest"]
    same = all(r[2]["digest"] == d0 for r in res)
    for r in res:
        st = r[2]["stats"]; n = max(st[0], 1)
        print("rank %d: loss %.5f  joins %s%s  timeouts %d  per step: %d keys requested, %d served, id blocks %d B (wire block %d words, full %d; %d full-size exchanges), "
              "rows %d B, gradients %d B, all-reduce %d B  (%.1f s)" % (
                  r[0], r[2]["loss"], "flags" if r[2]["join_mode"] == 1 else "events", "" if r[2]["join_mode"] == 1 else " (" + r[2]["why"] + ")", r[2]["timeouts"],
                  st[5] // n, st[6] // n, st[1] // n, st[7], st[9], st[8], st[2] // n, st[3] // n, st[4] // n, r[2]["seconds"]))
    print("%d ranks x %d steps at configs[2] size: replicated tensors %s across ranks" % (world, steps + 1, "bit-identical" if same else "DIFFER"))
    sys.exit(0 if same else 1)
