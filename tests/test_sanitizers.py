"""AddressSanitizer + UndefinedBehaviorSanitizer runs on the CPU (GPU sanitizers are not available on this pool):
the product's container reader on damaged files, and the oracle's encode/decode paths. Each runs in a subprocess."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, "tests", "golden", "examples")
FILES = [os.path.join(EX, n) for n in ("chord_cmajor_stereo.flo", "lossy_chord_high.flo", "telephone_8khz.flo",
                                       "audio_lossless.flo", "silence_1sec.flo", "lossy_music_pattern.flo")]

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")


def _libasan():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_product_container_reader_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "probe_fuzz")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-fno-omit-frame-pointer", os.path.join(ROOT, "tests", "native", "probe_fuzz.cpp"),
                           os.path.join(ROOT, "flo_amd", "csrc", "container.cpp"), "-o", exe])
    r = subprocess.run([exe] + FILES, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "accepted" in r.stdout


@pytest.mark.skipif(_libasan() is None, reason="libasan.so not found")
def test_oracle_under_asan_ubsan():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libflo_oracle_asan.so"])
    code = r'''
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, %r)
import oracle.oracle as O
O._SO_OVERRIDE = os.path.join(%r, "oracle", "libflo_oracle_asan.so")
rng = np.random.default_rng(3)
for ch in (1, 2, 3):
    pcm = (rng.uniform(-0.5, 0.5, 30000 * ch)).astype(np.float32)
    for q in (0.0, 0.55, 1.0):
        f = O.encode_lossy(pcm, 44100, ch, q)
        O.decode(f)
    for level in (0, 5, 9):
        f = O.encode_lossless(pcm, 44100, ch, 16, level)
        O.decode(f)
ex = %r
for name in sorted(os.listdir(ex)):
    if not name.endswith(".flo"):
        continue
    good = open(os.path.join(ex, name), "rb").read()
    O.decode(good)
    for i in range(60):
        bad = bytearray(good)
        for _ in range(1 + i %% 3):
            bad[int(rng.integers(0, len(bad)))] = int(rng.integers(0, 256))
        try:
            O.decode(bytes(bad))
        except RuntimeError:
            pass
        try:
            O.decode(bytes(bad[: int(rng.integers(0, len(bad)))]))
        except RuntimeError:
            pass
print("oracle asan ok")
''' % (ROOT, ROOT, EX)
    env = dict(os.environ, LD_PRELOAD=_libasan(), ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "oracle asan ok" in r.stdout, r.stdout[-2000:] + r.stderr[-6000:]
