"""The C-ABI library loads and exports every entry point include/radiomedium_hip.h declares.
No compute calls here (no GPU in the CPU test tier)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "radiomedium_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rm_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(rsa):
    L = C.CDLL(rsa.library_path())
    names = declared_symbols()
    assert len(names) >= 35
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_binding_covers_header(rsa):
    from radio_sim_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    assert _lib.lib().rm_abi_version() == 5


def test_struct_layouts(rsa):
    from radio_sim_amd import _lib
    assert C.sizeof(_lib.TxRecord) == 64
    assert C.sizeof(_lib.ModelParams) == 8 + 8 * 15 + 0  # 2 ints + 14 doubles + 1 uint64


def test_no_cpu_fallback_without_device(rsa):
    """Without a usable gfx950 device the product refuses to create a context."""
    from radio_sim_amd import _lib
    L = _lib.lib()
    n = L.rm_device_count()
    if n > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(rsa.RadioMediumError) as e:
        rsa.Engine(0)
    assert e.value.code == _lib.RM_ERR_NO_DEVICE
    assert b"no CPU fallback" in L.rm_last_error()


def test_product_does_not_touch_the_oracle():
    """Nothing under the product package or the C ABI may reference oracle/."""
    pkg = os.path.join(ROOT, "radio-sim_amd")
    offenders = []
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(base, f), errors="replace").read()
                if re.search(r"\boracle\b|rm_oracle|orc_", txt):
                    offenders.append(os.path.join(base, f))
    assert not offenders, offenders


def test_pure_helpers(rsa):
    from radio_sim_amd import _lib
    L = _lib.lib()
    assert L.rm_air_time_us(10) == 320                    # RadioPacket.java:72
    t0, t1 = C.c_int64(), C.c_int64()
    L.rm_event_times(1000, 320, 5000, C.byref(t0), C.byref(t1))
    assert (t0.value, t1.value) == (5000, 5320)           # Simulator.java:323-326
    p = _lib.ModelParams()
    L.rm_model_defaults(C.byref(p), 1)
    assert (p.udgm_transmission_range, p.udgm_interference_range, p.const_range) == (50.0, 100.0, 100.0)
    assert (p.udgm_success_ratio_rx, p.udgm_success_ratio_tx) == (1.0, 1.0)


def test_header_is_plain_c():
    """The boundary is a C ABI: the header must compile as C99 (and C++11) on its own, and a C caller links
    against the library's symbols."""
    import subprocess
    import tempfile
    hdr = os.path.join(ROOT, "include", "radiomedium_hip.h")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", hdr])
    subprocess.check_call(["g++", "-std=c++11", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", hdr])
    src = ('#include "radiomedium_hip.h"\n'
           "int main(void) { rm_model_params p; rm_model_defaults(&p, RM_MODEL_UDGM);\n"
           "  return (rm_abi_version() == RM_ABI_VERSION && rm_air_time_us(10) == 320 && p.udgm_transmission_range == 50.0) ? 0 : 1; }\n")
    with tempfile.TemporaryDirectory() as tmp:
        c_file, exe = os.path.join(tmp, "caller.c"), os.path.join(tmp, "caller")
        open(c_file, "w").write(src)
        lib = os.path.join(ROOT, "radio-sim_amd", "csrc")
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-I" + os.path.join(ROOT, "include"), c_file, "-L" + lib,
                               "-lradiomedium_hip", "-Wl,-rpath," + lib, "-o", exe])
        assert subprocess.run([exe], timeout=60).returncode == 0   # host-only entry points: no GPU needed


def test_jni_glue_type_checks_and_matches_the_java_declarations(tmp_path):
    """integration/jni/rm_jni.c and the `native` methods of the Java shim cannot be built here (no JDK).  The glue still
    goes through a C compiler's type checker -- against tests/cpp/jni_check/jni.h, the few JNI declarations it uses with
    the signatures of the JNI specification -- and every `private static native` method of GpuRadioMedium.java must have a
    C function of the mangled name with the same parameter list (jint / jlong / jdouble / arrays / ByteBuffer[])."""
    import re
    glue = os.path.join(ROOT, "integration", "jni", "rm_jni.c")
    obj = os.path.join(str(tmp_path), "rm_jni.o")
    subprocess.check_call(["gcc", "-std=c11", "-c", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "tests", "cpp", "jni_check"),
                           "-I" + os.path.join(ROOT, "include"), glue, "-o", obj])
    java = open(os.path.join(ROOT, "integration", "java", "se", "sics", "emul8", "radiomedium", "GpuRadioMedium.java")).read()
    csrc = re.sub(r"/\*.*?\*/", " ", open(glue).read(), flags=re.S)
    csrc = re.sub(r"//[^\n]*", " ", csrc)
    jtype = {"int": "jint", "long": "jlong", "double": "jdouble", "boolean": "jboolean", "String": "jstring", "int[]": "jintArray", "long[]": "jlongArray",
             "double[]": "jdoubleArray", "byte[]": "jbyteArray", "java.nio.ByteBuffer[]": "jobjectArray", "void": "void"}
    natives = re.findall(r"private static native\s+([\w.\[\]]+)\s+(\w+)\s*\(([^)]*)\)\s*;", re.sub(r"/\*.*?\*/", " ", java, flags=re.S))
    assert len(natives) >= 19
    cfuncs = {m.group(2): (m.group(1), m.group(3)) for m in re.finditer(r"JNIEXPORT\s+(\w+)\s+JFN\((\w+)\)\s*\(([^)]*)\)", csrc)}
    for ret, name, params in natives:
        assert name in cfuncs, "no C function for native method " + name
        cret, cparams = cfuncs[name]
        assert cret == jtype[ret], (name, ret, cret)
        want = [jtype[" ".join(p.split()[:-1])] for p in params.split(",") if p.strip()]
        got = [p.split()[0] for p in cparams.split(",")][2:]          # after JNIEnv *env, jclass cls
        assert got == want, (name, got, want)
    assert set(cfuncs) == {n for _, n, _ in natives}


def test_parity_harness_reads_the_goldens_as_they_are():
    """integration/java/harness/ReferenceParityHarness.java cannot be compiled here (no JDK); what can be held: the scenario
    files it names exist, the verdict code it compares with is the header's, and the .npy headers have the shape its reader
    expects (version 1, a quoted descr or a list of (name, type) pairs, C order)."""
    import zipfile
    src = open(os.path.join(ROOT, "integration", "java", "harness", "ReferenceParityHarness.java")).read()
    hdr = open(os.path.join(ROOT, "include", "radiomedium_hip.h")).read()
    delivered = int(re.search(r"RM_DELIVERED\s*=\s*(\d+)", hdr).group(1))
    assert "eVer.u8(i) == %d" % delivered in src
    names = re.findall(r'"(\w+\.npz)"', src)
    assert len(names) >= 5
    for name in names:
        z = zipfile.ZipFile(os.path.join(ROOT, "tests", "golden", name))
        for entry in z.namelist():
            raw = z.read(entry)
            assert raw[:6] == b"\x93NUMPY" and raw[6] == 1, entry
            header = raw[10:10 + (raw[8] | (raw[9] << 8))].decode("latin1")
            assert "'fortran_order': False" in header
            assert re.search(r"'descr':\s*'[<|][fiuU]\d+'", header) or re.findall(r"\('\w+',\s*'<[fi][48]'\)", header), (name, entry, header)
