// GpuModel.java -- drop-in model.Model over NativeKVStore: what Trainer/TrainerThread call.
// NOT compiled here (no JDK); shown for INTEGRATION.md.
package model;

import java.util.Map;
import org.jblas.FloatMatrix;
import store.NativeKVStore;
import update.Updater;

public class GpuModel implements Model {
    private final NativeKVStore kv;
    private final long handle;
    private final Map<String, Updater> updater;
    private final boolean wide;

    public GpuModel(NativeKVStore kv, long handle, Map<String, Updater> updater, boolean wide) {
        this.kv = kv; this.handle = handle; this.updater = updater; this.wide = wide;
        for (Map.Entry<String, Updater> e : updater.entrySet()) kv.setUpdater(e.getKey(), e.getValue().getName());
    }

    private static long[] ids(FloatMatrix m) {        // F x B column-major == [B][F] sample-major
        long[] out = new long[m.length];
        for (int i = 0; i < m.length; i++) out[i] = (long) m.data[i];
        return out;
    }

    @Override public float train(Map<String, FloatMatrix> datas) {
        FloatMatrix E = datas.get("E"), X = datas.get("X"), Y = datas.get("Y");
        return kv.train(handle, ids(E), X.data, wide ? ids(datas.get("W")) : null, Y.data, E.columns);
    }

    @Override public FloatMatrix predict(Map<String, FloatMatrix> datas) {
        FloatMatrix E = datas.get("E"), X = datas.get("X");
        float[] p = kv.predict(handle, ids(E), X.data, wide ? ids(datas.get("W")) : null, E.columns);
        return new FloatMatrix(1, p.length, p);
    }

    @Override public void pullWeights() { /* parameters stay resident in HBM behind the store */ }
    @Override public Map<String, Updater> getUpdater() { return updater; }
}
