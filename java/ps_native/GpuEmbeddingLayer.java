// GpuEmbeddingLayer.java -- drop-in layer.EmbeddingLayer on the GPU (layer/EmbeddingLayer.java:25-75): the per-field
// tables "emF<f>" live in HBM behind NativeKVStore; forward is one coalesced gather for all fields, backward is
// the per-key reduction + the (n+1)/(2n^2) factor of the reference's double backward + KVStore.sum + the "emF"
// updater fused into the scatter.  NOT compiled here (no JDK); shown for INTEGRATION.md.
package layer;

import org.jblas.FloatMatrix;
import store.NativeKVStore;

public class GpuEmbeddingLayer extends Layer {
    private final NativeKVStore kv;
    private final int F, D, ld;
    private int cap = 0;
    private long idsDev, aDev, deltaDev;
    private int deltaLd = 0;
    private boolean backwardDone = false;       // the reference runs backward() twice per step (SURVEY App. A.6)

    public GpuEmbeddingLayer(NativeKVStore kv, int embeddingFieldNum, int embeddingSize, long[] rowsPerField) {
        super("embedding", embeddingFieldNum, embeddingFieldNum * embeddingSize);
        this.kv = kv; this.F = embeddingFieldNum; this.D = embeddingSize;
        this.ld = (F * D + 15) / 16 * 16;
        kv.createEmbedding(rowsPerField, embeddingSize, 2, 0, 1, NativeKVStore.ROUTE_ID_MOD);
    }

    private void reserve(int B, int ldd) {
        if (B <= cap && ldd <= deltaLd) return;
        if (cap > 0) { kv.devFree(idsDev); kv.devFree(aDev); kv.devFree(deltaDev); }
        cap = Math.max(B, cap); deltaLd = Math.max(ldd, deltaLd);
        idsDev = kv.devAlloc(8L * cap * F); aDev = kv.devAlloc(4L * cap * ld); deltaDev = kv.devAlloc(4L * cap * deltaLd);
    }

    @Override public FloatMatrix forward() {
        FloatMatrix E = pre.A;                              // F x B, ids as floats (CTR.java:57)
        final int B = E.columns;
        reserve(B, ld);
        long[] ids = new long[B * F];
        for (int i = 0; i < ids.length; i++) ids[i] = (long) E.data[i];      // [B][F] sample-major == the matrix bytes
        kv.uploadLongs(idsDev, ids, ids.length);
        kv.embForward(idsDev, 0, B, NativeKVStore.ACT_RELU, aDev, ld);       // EmbeddingLayer.build: Relu (:53)
        float[] a = new float[B * ld];
        kv.downloadFloats(a, aDev, a.length);
        float[] out = new float[B * F * D];
        for (int b = 0; b < B; b++) System.arraycopy(a, b * ld, out, b * F * D, F * D);
        this.A = new FloatMatrix(F * D, B, out);
        return this.A;
    }

    @Override public FloatMatrix backward() {
        this.delta = next.delta;                            // (F*D [+X]) x B; the embedding columns come first
        if (backwardDone) return this.delta;                // second call of the step: its effect is inside grad mode COMPAT
        final int B = delta.columns, rows = delta.rows;
        reserve(B, rows);
        kv.uploadFloats(deltaDev, delta.data, B * rows);    // [B][rows]: ldd = rows
        kv.embBackwardUpdate(idsDev, 0, (long) B * F, B, NativeKVStore.ACT_RELU, aDev, ld, deltaDev, rows,
                             NativeKVStore.GRAD_COMPAT, NativeKVStore.SUM_AUTO, true);
        backwardDone = true;
        return this.delta;
    }

    /** the reference clears the per-field gradient maps here, once per step (layer/EmbeddingLayer.java:71-75) */
    @Override public void pullWeights() { backwardDone = false; }
}
