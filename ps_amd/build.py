"""Build libps_amd.so (gfx950 only) in-tree with hipcc.

    python -m ps_amd.build [--force]

One hipcc -c per translation unit (cached by mtime), then one link.  The
library is kept in-tree (ps_amd/lib/libps_amd.so): git-ignored, but it
travels to the GPU box with the gpurun snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libps_amd.so")
SOURCES = ["kernels_sort.hip", "kernels_gemm.hip", "kernels_emb.hip", "ps_store.hip", "ps_model.hip", "ps_ops.hip", "ps_layer_ops.hip", "ps_shard.hip", "ps_ingest.hip", "ps_eval.hip", "ps_ckpt.hip", "ps_comm.hip", "ps_keyed.hip"]
# -ffp-contract=off: the reference (JVM) rounds every float op separately;
# the updater / reduce kernels must too, to stay bit-exact with the oracle.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _newest_header():
    t = 0.0
    for d in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in os.listdir(d):
            if f.endswith((".h", ".inc")):
                t = max(t, os.path.getmtime(os.path.join(d, f)))
    return t


def _record_resources(obj_path, stderr):
    """hipcc's kernel-resource-usage remarks of one translation unit -> <obj>.resources.json ({kernel: {vgprs, scratch, ...}});
    everything else hipcc said goes to our stderr as usual.  tests/test_abi.py holds the hot kernels to zero scratch: an innocent
    edit (a second call site of an inlined function, round 5) once made hipcc keep a private copy of a kernel's argument struct --
    856 bytes of scratch per lane, the kernel eight times slower, every result still right."""
    import json, re
    res, cur, rest = {}, None, []
    keys = {"VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occupancy",
            "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill", "LDS Size [bytes/block]": "lds"}
    for line in stderr.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass-analysis=kernel-resource-usage\]", line)
        if not m:
            rest.append(line)
            continue
        body = m.group(1).strip()
        if body.startswith("Function Name:"):
            cur = body.split(":", 1)[1].strip()
            res[cur] = {}
        elif cur is not None and ":" in body:
            k, v = body.rsplit(":", 1)
            if k.strip() in keys:
                res[cur][keys[k.strip()]] = int(v)
    if rest:
        sys.stderr.write("\n".join(rest) + "\n")
    with open(obj_path + ".resources.json", "w") as f:
        json.dump(res, f, indent=0, sort_keys=True)


def kernel_resources():
    """{mangled kernel name: {vgprs, scratch, occupancy, ...}} of the library as last compiled (build() first)."""
    import json
    out = {}
    for f in sorted(os.listdir(OBJ)):
        if f.endswith(".resources.json"):
            out.update(json.load(open(os.path.join(OBJ, f))))
    return out


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    hdr_t = _newest_header()
    objs, relink, todo = [], force or not os.path.exists(LIB), []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        op = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(op)
        if force or not os.path.exists(op) or not os.path.exists(op + ".resources.json") or os.path.getmtime(op) < max(os.path.getmtime(sp), hdr_t):
            todo.append((sp, op))

    def compile_one(job):
        sp, op = job
        cmd = [hipcc] + FLAGS + ["-Rpass-analysis=kernel-resource-usage"] + os.environ.get("PS_AMD_EXTRA_FLAGS", "").split() + ["-c", sp, "-o", op]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        _record_resources(op, r.stderr)
        if r.returncode != 0:
            raise subprocess.CalledProcessError(r.returncode, cmd)

    if todo:        # the translation units are independent: one hipcc per core
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=max(1, min(len(todo), os.cpu_count() or 1))) as ex:
            list(ex.map(compile_one, todo))
        relink = True
    if relink or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
