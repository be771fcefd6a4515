"""The JNI shim (java/ps_native/ps_jni.cpp) cannot be linked here (no JDK in the image).  What CAN be checked:
it is valid C++ against the JNI signatures it uses (g++ -fsyntax-only with the declaration-only tests/jni_mock/jni.h
and the real include/ps_native.h, so every C-ABI call in it type-checks), and every `native` method declared in
NativeKVStore.java has exactly one Java_store_NativeKVStore_* definition with the same number of parameters."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JAVA = os.path.join(ROOT, "java", "ps_native")


def test_jni_shim_is_valid_cpp_against_the_abi_header():
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "tests", "jni_mock"),
           "-I" + os.path.join(ROOT, "include"), os.path.join(JAVA, "ps_jni.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]


def _split_params(s):
    s = s.strip()
    return [] if not s else [p for p in s.split(",")]


def test_every_native_method_has_its_definition():
    java = open(os.path.join(JAVA, "NativeKVStore.java")).read()
    cpp = open(os.path.join(JAVA, "ps_jni.cpp")).read()
    natives = re.findall(r"\bnative\s+[\w\[\]]+\s+(\w+)\s*\(([^)]*)\)\s*;", java)
    assert len(natives) >= 30
    defs = {m.group(1): m.group(2) for m in re.finditer(r"Java_store_NativeKVStore_(\w+)\s*\(([^)]*)\)", cpp)}
    for name, params in natives:
        assert name in defs, "native %s has no definition in ps_jni.cpp" % name
        njava = len(_split_params(params))
        ncpp = len(_split_params(defs[name])) - 2                      # JNIEnv*, jobject/jclass
        assert njava == ncpp, "%s: %d Java parameters, %d in the shim" % (name, njava, ncpp)
    assert set(defs) == {n for n, _ in natives}, "definitions without a declaration: %s" % (set(defs) - {n for n, _ in natives})


def test_layer_classes_only_call_declared_natives():
    java = open(os.path.join(JAVA, "NativeKVStore.java")).read()
    methods = set(re.findall(r"\bpublic\s+(?:static\s+)?(?:native\s+)?[\w\[\]]+\s+(\w+)\s*\(", java))
    for f in ("GpuFcLayer.java", "GpuEmbeddingLayer.java", "GpuModel.java"):
        src = open(os.path.join(JAVA, f)).read()
        for call in re.findall(r"\bkv\.(\w+)\s*\(", src):
            assert call in methods, "%s calls kv.%s, which NativeKVStore does not declare" % (f, call)
