"""The copy threads of the staged host path (steroscopic-hardware_amd/host/copy_gang.h: plain C++, no HIP) under
ThreadSanitizer and AddressSanitizer + UBSan on the CPU (GPU sanitizers are not available on the pool): the hand-off
between the calling thread and the helpers (spin phase, sleep phase, shutdown), and the bounds of every copy -- sources
allocated at exactly the bytes a copy may read, destinations checked for bytes outside their rows."""
import os
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "tests", "host", "stress_copy_gang.cpp")
HDR = os.path.join(REPO, "steroscopic-hardware_amd", "host", "copy_gang.h")
GRAY_SRC = os.path.join(REPO, "tests", "host", "check_gray_rows.cpp")
GRAY_HDR = os.path.join(REPO, "steroscopic-hardware_amd", "host", "gray_rows.h")
OUT = os.path.join(REPO, "steroscopic-hardware_amd", "lib")


def _build(tag, flags):
    exe = os.path.join(OUT, "copy_gang_" + tag)
    if not os.path.exists(exe) or max(os.path.getmtime(SRC), os.path.getmtime(HDR), os.path.getmtime(GRAY_HDR)) > os.path.getmtime(exe):
        os.makedirs(OUT, exist_ok=True)
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-pthread", "-Wall", "-Wextra", "-Werror"] + flags + ["-o", exe, SRC])
    return exe


@pytest.mark.parametrize("tag,flags,env,marker", [
    ("tsan", ["-fsanitize=thread"], {"TSAN_OPTIONS": "halt_on_error=1 exitcode=66"}, "ThreadSanitizer"),
    ("asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"],
     {"ASAN_OPTIONS": "detect_leaks=1 halt_on_error=1 exitcode=67"}, "AddressSanitizer"),
])
def test_copy_gang_under_sanitizers(tag, flags, env, marker):
    r = subprocess.run([_build(tag, flags), "100"], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "OK gangs=52" in r.stdout and "violations=0" in r.stdout, r.stdout
    assert marker not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]


def test_host_gray_rows_match_the_go_formula_under_asan():
    """SADHIP_FLAG_HOST_GRAY: the vector luminance of host/gray_rows.h equals the scalar formula (gray.go:52-55 on
    color.NRGBA.RGBA()) for every width, alpha class and channel count, reads nothing past a row (sources allocated at exactly
    w * channels bytes under AddressSanitizer), writes nothing past w, alone and split over the copy threads."""
    exe = os.path.join(OUT, "check_gray_rows_asan")
    if not os.path.exists(exe) or max(os.path.getmtime(GRAY_SRC), os.path.getmtime(HDR), os.path.getmtime(GRAY_HDR)) > os.path.getmtime(exe):
        os.makedirs(OUT, exist_ok=True)
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-g", "-pthread", "-Wall", "-Wextra", "-Werror", "-fsanitize=address,undefined",
                               "-fno-sanitize-recover=undefined", "-o", exe, GRAY_SRC])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1 halt_on_error=1 exitcode=67"))
    assert r.returncode == 0 and "gray rows: bad=0" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
