// GpuFcLayer.java -- drop-in layer.FcLayer on the GPU: same seams (layer/Layer.java:12-78: forward(), backward(),
// pullWeights(); inputs through pre.A, gradients through next.delta or its own delta when last,
// layer/FcLayer.java:95-99), weights "fc<i>.weights" / "fc<i>.bias" resident in HBM behind NativeKVStore.
// NOT compiled here (no JDK); shown for INTEGRATION.md.
//
// A FloatMatrix "features x B" (column-major) is byte-for-byte a row-major [B][features] array, which is the
// layout of the device buffers: forward packs pre.A into [B][ldx] (ldx = in+1 rounded up to 16, a ones column at
// [in] carries the bias through the GEMM), backward uploads the incoming delta into [B][ldd] (ldd = out rounded up
// to 16).  Two host<->device copies per call keep the class usable between UNMODIFIED reference layers; a chain of
// Gpu*Layer objects hands device buffers over instead (devA / devDelta below) and never touches the host.
package layer;

import org.jblas.FloatMatrix;
import store.NativeKVStore;

public class GpuFcLayer extends Layer {
    private final NativeKVStore kv;
    private final int index, act, ldx, ldd;
    private int cap = 0;                       // batch capacity of the device buffers
    private long xDev, yDev, deltaDev, dxDev;  // [B][ldx], [B][ldd], [B][ldd], [B][ldxPrev]
    long devA, devDelta;                       // device views for a neighbouring Gpu*Layer (0: go through the host)

    /** index i of "fc<i>"; act: NativeKVStore.ACT_RELU (hidden), ACT_SIGMOID (last DNN layer), ACT_NONE (WideDeep's last) */
    public GpuFcLayer(NativeKVStore kv, int index, int inputDims, int outputDims, int act) {
        super("fc" + index, inputDims, outputDims);
        this.kv = kv; this.index = index; this.act = act;
        this.ldx = (inputDims + 1 + 15) / 16 * 16;
        this.ldd = (outputDims + 15) / 16 * 16;
        kv.createFc(index, inputDims, outputDims);          // lazily creates the tensors like kvStore.get(key, init)
    }

    private void reserve(int B) {
        if (B <= cap) return;
        if (cap > 0) { kv.devFree(xDev); kv.devFree(yDev); kv.devFree(deltaDev); kv.devFree(dxDev); }
        xDev = kv.devAlloc(4L * B * ldx); yDev = kv.devAlloc(4L * B * ldd);
        deltaDev = kv.devAlloc(4L * B * ldd); dxDev = kv.devAlloc(4L * B * ((inputDims + 15) / 16 * 16));
        cap = B;
    }

    @Override public FloatMatrix forward() {
        FloatMatrix in = this.pre.A;                        // inputDims x B
        final int B = in.columns;
        reserve(B);
        float[] x = new float[B * ldx];
        for (int b = 0; b < B; b++) {
            System.arraycopy(in.data, b * inputDims, x, b * ldx, inputDims);
            x[b * ldx + inputDims] = 1f;                    // the bias column
        }
        kv.uploadFloats(xDev, x, x.length);
        kv.fcForward(index, act, xDev, ldx, B, yDev, ldd);
        float[] y = new float[B * ldd];
        kv.downloadFloats(y, yDev, y.length);
        float[] a = new float[B * outputDims];
        for (int b = 0; b < B; b++) System.arraycopy(y, b * ldd, a, b * outputDims, outputDims);
        this.A = new FloatMatrix(outputDims, B, a);         // Z == A afterwards, as in the reference
        this.devA = yDev;
        return this.A;
    }

    @Override public FloatMatrix backward() {
        FloatMatrix d = this.next == null ? this.delta : this.next.delta;   // layer/FcLayer.java:95-99
        final int B = d.columns;
        float[] dl = new float[B * ldd];
        for (int b = 0; b < B; b++) System.arraycopy(d.data, b * outputDims, dl, b * ldd, outputDims);
        kv.uploadFloats(deltaDev, dl, dl.length);
        final int lddx = (inputDims + 15) / 16 * 16;
        // act' in place, db and dW into the store's pending sums (kvStore.sum), W^T delta out
        kv.fcBackward(index, act, xDev, ldx, yDev, ldd, deltaDev, ldd, B, dxDev, lddx);
        float[] dx = new float[B * lddx];
        kv.downloadFloats(dx, dxDev, dx.length);
        float[] out = new float[B * inputDims];
        for (int b = 0; b < B; b++) System.arraycopy(dx, b * lddx, out, b * inputDims, inputDims);
        this.delta = new FloatMatrix(inputDims, B, out);
        this.devDelta = dxDev;
        return this.delta;
    }

    /** Trainer calls kvStore.update(updaters) after backward (train/Trainer.java:90-100): the dense half of it */
    public void update() { kv.denseUpdate(index); }

    @Override public void pullWeights() { /* "fc<i>.weights" stays resident in HBM */ }
}
