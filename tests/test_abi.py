"""CPU-side checks of the drop-in boundary: the library loads, exports every symbol include/flo_hip.h declares,
and refuses to run without a GPU (no silent CPU fallback)."""
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "flo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(flo_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported():
    from flo_amd import _native
    L = _native.lib()
    declared = _declared_symbols()
    assert len(declared) >= 25
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(_native.EXPORTS) == declared


def test_product_does_not_reference_the_oracle():
    # the oracle is test infrastructure: nothing under flo_amd/ may import, link or dlopen it
    for dirpath, _, files in os.walk(os.path.join(ROOT, "flo_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "flo_oracle" not in txt and "oracle/" not in txt.replace("oracle/ is test", ""), (dirpath, f)
    import subprocess
    out = subprocess.run(["ldd", os.path.join(ROOT, "flo_amd", "libflo_hip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out and "amdhip64" in out
    assert "librccl" in out        # the multi-GPU exchange step (flo_dist_*) is written against RCCL directly


def test_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import flo_amd
    with pytest.raises(flo_amd.FloError) as e:
        flo_amd.Context(0)
    assert "no CPU fallback" in str(e.value) or "HIP" in str(e.value)
    with pytest.raises(flo_amd.FloError):
        flo_amd.encode_lossy([0.0] * 100, 44100, 1, 16, 2)


def test_quality_preset_mirror():
    # libflo/tests/rust/lossy_quality_tests.rs:5-28
    from flo_amd import QualityPreset as Q
    assert Q.Low.as_f32() == 0.0 and Q.Transparent.as_f32() == 1.0
    for i in range(5):
        assert int(Q(i)) == i and Q.from_f32(Q(i).as_f32()) == Q(i)
    assert Q.from_bitrate(128, 44100, 2) == Q.Medium and Q.from_bitrate(320, 44100, 2) == Q.VeryHigh
    assert Q.from_bitrate(48, 44100, 2) == Q.Low and Q.from_bitrate(400, 44100, 2) == Q.Transparent


def test_public_headers_compile_as_plain_c99(tmp_path):
    # the boundary is a C ABI: the headers must be usable from C (and from the C subset a Rust bindgen run sees)
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("needs gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "t.c"
    src.write_text('#include "flo_hip.h"\n#include "flo_synth.h"\nint main(void) { return FLO_OK; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"),
                           "-fsyntax-only", str(src)])


def _build_native_caller(tmp_path):
    """tests/native/abi_smoke.c linked against libflo_hip.so with gcc: a caller with no Python in the loop."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("needs gcc")
    exe = tmp_path / "abi_smoke"
    libdir = os.path.join(ROOT, "flo_amd")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "native", "abi_smoke.c"), "-o", str(exe),
                           "-L", libdir, "-lflo_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath-link,/opt/rocm/lib",
                           "-Wl,-rpath,/opt/rocm/lib"])
    gold = os.path.join(ROOT, "tests", "golden", "examples")
    return [str(exe), os.path.join(gold, "audio_lossless.flo"), os.path.join(gold, "audio_lossy.flo")]


def test_native_c_caller_links_and_fails_loudly_without_gpu(tmp_path):
    import subprocess
    import torch
    cmd = _build_native_caller(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: the run itself is test_native_c_caller_reproduces_the_reference_files")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stderr)
    assert "no HIP device" in r.stderr or "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_native_c_caller_reproduces_the_reference_files(tmp_path):
    # flo_encode_lossless / flo_encode_lossy / flo_decode driven from C on BASELINE configs[0]'s input: header + TOC +
    # DATA of the lossless file and the DATA chunk of the lossy file equal the files the reference wrote
    import subprocess
    r = subprocess.run(_build_native_caller(tmp_path), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "abi_smoke ok" in r.stdout
