// ps_ckpt.hip -- shard checkpoint / resume.
//
// The reference has no checkpoint; its wire format for a parameter is Matrix{key,row,cols,data}
// (ps.proto:16-23).  A GPU shard's state is a handful of large dense arrays, so the file is those arrays
// raw (f32, the store's HBM layout), behind a small header that pins what they mean:
//
//   magic "PSAMDCK1", header (device-independent ints), then sections in fixed order:
//     embedding W [total_rows][D], state [total_rows][2][D] (if any)
//     wide W [rows], state [rows][2], touched [rows] (u8), bias[1], bias_state[2]      (if a wide table exists)
//     per FC layer: W' [Kpad][ldw], S1, S2                                            (Wt is rebuilt on load)
//     the updaters registered on the store (name -> parameters), globalStep
//
// One file per shard (rank): every rank saves its own rows, the replicated tensors are identical on all ranks.
// Load requires a store created with the same geometry (fields, vocabularies, D, shard/nshards, wide size, FC
// shapes); anything else is PS_E_BAD_ARG.  Resume is exact: train k steps, save, load into a fresh store, and
// the following steps are bit-identical to the uninterrupted run (tests/test_gpu_ckpt.py).
#include <stdio.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <string>
#include <vector>

#include "ps_store.h"

namespace {

const char kMagic[8] = {'P', 'S', 'A', 'M', 'D', 'C', 'K', '1'};

struct Hdr {
    int64_t seed, global_step, emb_total_rows, wide_rows;
    int32_t F, D, state_slots, shard, nshards, route_mode, nfc, nupd;
};

struct IO {
    FILE *f = nullptr;
    ps_store *s = nullptr;
    std::vector<char> buf;       // staging (the arrays go through the host in 64 MiB pieces)
    bool write = false;
    bool ok = true;

    void raw(void *p, size_t n) {
        if (!ok || n == 0) return;
        const size_t r = write ? fwrite(p, 1, n, f) : fread(p, 1, n, f);
        if (r != n) ok = false;
    }
    // device array <-> file
    int dev(void *dptr, size_t bytes) {
        const size_t piece = (size_t)64 << 20;
        if (buf.size() < (bytes < piece ? bytes : piece)) buf.resize(bytes < piece ? bytes : piece);
        for (size_t off = 0; off < bytes && ok; off += piece) {
            const size_t n = bytes - off < piece ? bytes - off : piece;
            if (write) {
                HIPCHK(hipMemcpyAsync(buf.data(), (char *)dptr + off, n, hipMemcpyDeviceToHost, s->stream));
                HIPCHK(hipStreamSynchronize(s->stream));
                raw(buf.data(), n);
            } else {
                raw(buf.data(), n);
                if (!ok) break;
                HIPCHK(hipMemcpyAsync((char *)dptr + off, buf.data(), n, hipMemcpyHostToDevice, s->stream));
                HIPCHK(hipStreamSynchronize(s->stream));
            }
        }
        return PS_OK;
    }
};

// bytes sections() moves for this store's geometry (what a complete file holds after its header)
size_t section_bytes(const ps_store *s) {
    size_t n = 0;
    const EmbTables &e = s->emb;
    if (e.W) {
        n += sizeof(float) * (size_t)e.total_rows * e.D;
        if (e.state) n += sizeof(float) * (size_t)e.total_rows * 2 * e.D;
    }
    const WideTable &w = s->wide;
    if (w.W) n += sizeof(float) * (size_t)w.rows * 3 + (size_t)w.rows + sizeof(float) * 3;
    for (auto &f : s->fc)
        if (f.present) n += 3 * sizeof(float) * (size_t)f.Kpad * f.ldw;
    return n;
}

const char kEnd[8] = {'P', 'S', 'A', 'M', 'D', 'E', 'N', 'D'};

int sections(IO &io) {
    ps_store *s = io.s;
    EmbTables &e = s->emb;
    if (e.W) {
        PSCHK(io.dev(e.W, sizeof(float) * (size_t)e.total_rows * e.D));
        if (e.state) PSCHK(io.dev(e.state, sizeof(float) * (size_t)e.total_rows * 2 * e.D));
    }
    WideTable &w = s->wide;
    if (w.W) {
        PSCHK(io.dev(w.W, sizeof(float) * (size_t)w.rows));
        PSCHK(io.dev(w.state, sizeof(float) * (size_t)w.rows * 2));
        PSCHK(io.dev(w.touched, (size_t)w.rows));
        PSCHK(io.dev(w.bias, sizeof(float)));
        PSCHK(io.dev(w.bias_state, sizeof(float) * 2));
    }
    for (auto &f : s->fc) {
        if (!f.present) continue;
        const size_t n = sizeof(float) * (size_t)f.Kpad * f.ldw;
        PSCHK(io.dev(f.W, n));
        PSCHK(io.dev(f.S1, n));
        PSCHK(io.dev(f.S2, n));
    }
    return PS_OK;
}

}  // namespace

// defined in kernels_emb.hip: Wt = W'^T after a load
int launch_transpose_w(const float *W, float *Wt, int Kpad, int ldw, int N, hipStream_t st);

extern "C" int ps_store_save(ps_store_t *s, const char *path) {
    if (!s || !path) return ps_set_err(PS_E_BAD_ARG, "null argument");
    PSCHK(store_enter(s));
    HIPCHK(hipStreamSynchronize(s->stream));
    IO io;
    io.s = s; io.write = true;
    // written beside the target and renamed over it once complete and on disk: a crash (or a full disk) in the middle
    // of a save never destroys the previous checkpoint
    const std::string tmp = std::string(path) + ".tmp";
    io.f = fopen(tmp.c_str(), "wb");
    if (!io.f) return ps_set_err(PS_MISSING, "cannot create %s", tmp.c_str());
    Hdr h;
    memset(&h, 0, sizeof h);
    h.seed = (int64_t)s->seed; h.global_step = s->global_step;
    h.emb_total_rows = s->emb.W ? s->emb.total_rows : 0; h.wide_rows = s->wide.W ? s->wide.rows : 0;
    h.F = s->emb.F; h.D = s->emb.D; h.state_slots = s->emb.state ? 2 : 0;
    h.shard = s->emb.shard; h.nshards = s->emb.nshards; h.route_mode = s->emb.route_mode;
    h.nfc = 0;
    for (auto &f : s->fc) h.nfc += f.present ? 1 : 0;
    h.nupd = (int32_t)s->updaters.size();
    io.raw((void *)kMagic, 8);
    io.raw(&h, sizeof h);
    for (int f = 0; f < s->emb.F; ++f) { int64_t v = s->emb.rows[f]; io.raw(&v, 8); }
    for (size_t l = 0; l < s->fc.size(); ++l) {
        if (!s->fc[l].present) continue;
        int32_t d[3] = {(int32_t)l, s->fc[l].K, s->fc[l].N};
        io.raw(d, sizeof d);
    }
    for (auto &kv : s->updaters) {
        int32_t n = (int32_t)kv.first.size();
        io.raw(&n, 4);
        io.raw((void *)kv.first.data(), (size_t)n);
        ps_updater_t u = kv.second;
        io.raw(&u, sizeof u);
    }
    int rc = sections(io);
    io.raw((void *)kEnd, 8);
    bool ok = io.ok && fflush(io.f) == 0 && fsync(fileno(io.f)) == 0;
    ok = (fclose(io.f) == 0) && ok;
    if (rc != PS_OK || !ok) {
        (void)remove(tmp.c_str());
        return rc != PS_OK ? rc : ps_set_err(PS_E_HIP, "short write to %s", tmp.c_str());
    }
    if (rename(tmp.c_str(), path) != 0) { (void)remove(tmp.c_str()); return ps_set_err(PS_E_HIP, "cannot rename %s to %s", tmp.c_str(), path); }
    return PS_OK;
}

extern "C" int ps_store_load(ps_store_t *s, const char *path) {
    if (!s || !path) return ps_set_err(PS_E_BAD_ARG, "null argument");
    PSCHK(store_enter(s));
    HIPCHK(hipStreamSynchronize(s->stream));
    IO io;
    io.s = s; io.write = false;
    io.f = fopen(path, "rb");
    if (!io.f) return ps_set_err(PS_MISSING, "cannot open %s", path);
    auto fail = [&](int code, const char *what) { fclose(io.f); return ps_set_err(code, "%s: %s", path, what); };
    char magic[8];
    Hdr h;
    io.raw(magic, 8);
    io.raw(&h, sizeof h);
    if (!io.ok || memcmp(magic, kMagic, 8) != 0) return fail(PS_E_BAD_ARG, "not a ps_amd checkpoint");
    const int64_t have_emb = s->emb.W ? s->emb.total_rows : 0, have_wide = s->wide.W ? s->wide.rows : 0;
    if (h.F != s->emb.F || h.D != s->emb.D || h.emb_total_rows != have_emb || h.shard != s->emb.shard ||
        h.nshards != s->emb.nshards || h.route_mode != s->emb.route_mode || h.state_slots != (s->emb.state ? 2 : 0))
        return fail(PS_E_BAD_ARG, "embedding geometry differs from this store (fields, D, rows, shard/nshards, state)");
    if (h.wide_rows != have_wide) return fail(PS_E_BAD_ARG, "wide table size differs from this store");
    for (int f = 0; f < h.F; ++f) {
        int64_t v = 0;
        io.raw(&v, 8);
        if (!io.ok || v != s->emb.rows[f]) return fail(PS_E_BAD_ARG, "a field's vocabulary differs from this store");
    }
    int nfc_here = 0;
    for (auto &f : s->fc) nfc_here += f.present ? 1 : 0;
    if (h.nfc != nfc_here) return fail(PS_E_BAD_ARG, "number of FC layers differs from this store");
    for (int i = 0; i < h.nfc; ++i) {
        int32_t d[3];
        io.raw(d, sizeof d);
        if (!io.ok || d[0] < 0 || d[0] >= (int)s->fc.size() || !s->fc[d[0]].present || s->fc[d[0]].K != d[1] || s->fc[d[0]].N != d[2])
            return fail(PS_E_BAD_ARG, "an FC layer's shape differs from this store");
    }
    std::map<std::string, ps_updater_t> upd;
    for (int i = 0; i < h.nupd; ++i) {
        int32_t n = 0;
        io.raw(&n, 4);
        if (!io.ok || n < 0 || n > 4096) return fail(PS_E_BAD_ARG, "corrupt updater table");
        std::string key((size_t)n, '\0');
        io.raw(&key[0], (size_t)n);
        ps_updater_t u;
        io.raw(&u, sizeof u);
        if (!io.ok) return fail(PS_E_BAD_ARG, "corrupt updater table");
        upd[key] = u;
    }
    // Nothing on the device has been touched so far.  The header fixed the geometry, so the exact length of a complete
    // file is known: check it (and the end marker) BEFORE the first byte goes to HBM -- a truncated or padded file is
    // refused with the store untouched instead of half overwritten.
    {
        struct stat stt;
        const long pos = ftell(io.f);
        if (pos < 0 || fstat(fileno(io.f), &stt) != 0) return fail(PS_E_HIP, "cannot stat");
        const uint64_t want = (uint64_t)pos + (uint64_t)section_bytes(s) + 8;
        if ((uint64_t)stt.st_size != want) return fail(PS_E_BAD_ARG, "truncated or oversized checkpoint (the store was not modified)");
        char endm[8];
        if (fseek(io.f, -8, SEEK_END) != 0 || fread(endm, 1, 8, io.f) != 8 || memcmp(endm, kEnd, 8) != 0 || fseek(io.f, pos, SEEK_SET) != 0)
            return fail(PS_E_BAD_ARG, "end marker missing: incomplete checkpoint (the store was not modified)");
    }
    int rc = sections(io);
    fclose(io.f);
    if (rc != PS_OK || !io.ok)
        return ps_set_err(rc != PS_OK ? rc : PS_E_HIP, "%s: read error while loading; the store now holds a MIX of old and new tensors -- reload or recreate it", path);
    for (auto &f : s->fc)
        if (f.present) { PSCHK(launch_transpose_w(f.W, f.Wt, f.Kpad, f.ldw, f.N, s->stream)); PSCHK(launch_pack_w(f.Wt, f.Wp, f.N, f.Kpad, s->stream)); }
    HIPCHK(hipStreamSynchronize(s->stream));
    s->updaters = upd;
    s->global_step = h.global_step;
    return PS_OK;
}
