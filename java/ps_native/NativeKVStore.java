// NativeKVStore.java -- the reference-side binding of libps_amd.so (include/ps_native.h).
// NOT compiled in this repository's image (no JDK / jni.h here); it is the stub a maintainer of
// wudikua/ps adds next to store/KVStore.java.  One instance per GPU shard.  Every `native` below has its
// definition in ps_jni.cpp (tests/test_jni_shim.py checks the two lists against each other).
package store;

public final class NativeKVStore implements AutoCloseable {
    static { System.loadLibrary("ps_amd_jni"); }       // ps_jni.cpp, linked against libps_amd.so

    // constants of include/ps_native.h
    public static final int ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID = 2;
    public static final int GRAD_COMPAT = 0, GRAD_INTENDED = 1;
    public static final int SUM_AUTO = 0, SUM_SEQUENTIAL = 1, SUM_CHUNKED = 2;
    public static final int ROUTE_ID_MOD = 0, ROUTE_JAVA_STRING = 1;
    public static final int MODEL_DNN = 0, MODEL_WIDEDEEP = 1;

    private long handle;                                // ps_store_t*

    public NativeKVStore(int device, long seed) { handle = create(device, seed); }

    // ---- store.KVStore (store/KVStore.java:129-166) ------------------------------------------
    /** KVStore.get(key): null when absent (Resp 204). Keys: "emF3.28305.0", "fc0.weights", "wide.bias", ... */
    public native float[] get(String key);
    /** KVStore.put(key, val) */
    public native void put(String key, float[] val);
    /** PSClient.getList / updateList for one field (net/PSClient.java:72-98,128-151); dim = embedding size */
    public native float[] getRows(int field, long[] ids, int which, int dim);
    public native void putRows(int field, long[] ids, int which, float[] rows);
    /** Map<String,Updater>.put(key, updater) by Updater.getName() string (update/AdamUpdater.java:72-74) */
    public native void setUpdater(String keyOrPrefix, String updaterName);
    public native long globalStep();

    // ---- tables -----------------------------------------------------------------------------------
    /** routeMode ROUTE_JAVA_STRING = net/Mod.java:13-15 (String.hashCode(key) mod n, floorMod) */
    public native void createEmbedding(long[] rowsPerField, int dim, int stateSlots, int shard, int nshards, int routeMode);
    public native void createWide(long wideSize);
    public native void createFc(int layer, int in, int out);

    // ---- model/DNN.java, model/WideDeepNN.java: the whole step in one call -----------------------------
    /** buildModel(...): returns a ps_model_t* handle */
    public native long buildModel(int kind, int F, int D, int X, int[] fcDims, long wideSize, int maxBatch);
    /** TrainerThread.call + KVStore.update + clear for thread = 1; E is the F x B id matrix as long[B*F]
     *  (sample-major = the bytes of the column-major FloatMatrix), X [B*numberFieldNum], W wide ids, Y labels. */
    public native float train(long model, long[] E, float[] X, long[] W, float[] Y, int B);
    public native float[] predict(long model, long[] E, float[] X, long[] W, int B);
    public native void destroyModel(long model);

    // ---- -Dmode=dist: one process per GPU is worker and owner (net/PSRouterClient.java:60-151, net/PServer.java:102-283)
    public static native byte[] commUniqueId();                                   // rank 0; hand the 256 bytes to every rank
    public native long commCreate(int nranks, int rank, byte[] id);               // RCCL inside libps_amd (ps_comm_rccl_create)
    public native void commSelfCheck(long comm);                                  // every collective once on known patterns
    public native void commDestroy(long comm);
    public native float shardStep(long model, long comm, long[] E, float[] X, long[] W, float[] Y, int B, boolean isPsAsync);

    // ---- layer.Layer granularity (GpuEmbeddingLayer / GpuFcLayer): device buffers are long handles ---------
    public native long devAlloc(long bytes);
    public native void devFree(long dev);
    public native void uploadFloats(long dstDev, float[] src, int n);
    public native void uploadLongs(long dstDev, long[] src, int n);
    public native void downloadFloats(float[] dst, long srcDev, int n);
    /** EmbeddingLayer.forward (layer/EmbeddingLayer.java:25-48) for all fields */
    public native void embForward(long idsDev, long offsetsDev, int B, int act, long outDev, int ld);
    /** EmbeddingLayer.backward x2 + KVStore.sum + update fused (layer/EmbeddingField.java:86-104) */
    public native void embBackwardUpdate(long idsDev, long offsetsDev, long nnz, int B, int act, long aDev, int lda,
                                         long deltaDev, int ldd, int gradMode, int sumOrder, boolean apply);
    /** FcLayer.forward / backward (layer/FcLayer.java:74-110); backward leaves dW, db in the store's pending sums */
    public native void fcForward(int layer, int act, long xDev, int ldx, int B, long yDev, int ldy);
    public native void fcBackward(int layer, int act, long xDev, int ldx, long yDev, int ldy, long deltaDev, int ldd, int B, long dxDev, int lddx);
    /** KVStore.update(Map) + clear for the dense tensors; layer < 0: all (store/KVStore.java:240-277) */
    public native void denseUpdate(int layer);
    public native void sync();

    // ---- the rows either side of the path --------------------------------------------------------------
    public native double auc(float[] p, float[] y);                               // evaluate/AUC.java
    public native void save(String path);                                         // shard checkpoint
    public native void load(String path);

    private static native long create(int device, long seed);
    private static native void destroy(long handle);
    long handle() { return handle; }
    @Override public void close() { if (handle != 0) { destroy(handle); handle = 0; } }
}
