"""Host-side sanitizer tier for the C++ runtime: csrc/runtime/rccl_transport.cpp
is compiled against mock HIP/RCCL headers (tests/native/mock) with AddressSanitizer + UBSan and
driven by tests/native/transport_host_test.cpp — no GPU, no RCCL library needed; csrc/runtime/pinned_ring.cpp
runs under ThreadSanitizer against an asynchronous mock (tests/native/mock_async, ring_tsan_test.cpp)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs a host C++ compiler")
def test_rccl_transport_host_logic_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "transport_host_test")
    mock = os.path.join(ROOT, "tests", "native", "mock")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=all", "-static-libasan", "-I", mock, "-I", os.path.join(ROOT, "tests", "native"),
           os.path.join(ROOT, "distributed_training_pytorch_amd", "csrc", "runtime", "rccl_transport.cpp"),
           os.path.join(ROOT, "tests", "native", "transport_host_test.cpp"), "-o", exe, "-lpthread", "-ldl"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-4000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:verify_asan_link_order=0")
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120, env=env)
    assert r.returncode == 0 and "transport host test ok" in r.stdout, r.stdout[-4000:]


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs a host C++ compiler")
def test_pinned_ring_protocol_under_thread_sanitizer(tmp_path):
    """csrc/runtime/pinned_ring.cpp under ThreadSanitizer against an ASYNCHRONOUS mock of HIP (one worker
    thread per stream: the copy stream really reads the pinned slots while producer threads refill others and
    the compute stream reads the device buffers), driven with the protocol of data/prefetch.py's refilling ring
    source.  40 batches through 3 slots arrive intact and in order, no data race, nothing leaked.  Negative
    control: the same program without the producer's host_wait() must be REPORTED by TSan — the detector sees
    the hand-over it is there to guard."""
    mock = os.path.join(ROOT, "tests", "native", "mock_async")
    base = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=thread", "-I", mock,
            os.path.join(ROOT, "distributed_training_pytorch_amd", "csrc", "runtime", "pinned_ring.cpp"),
            os.path.join(ROOT, "tests", "native", "ring_tsan_test.cpp"), "-lpthread"]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1:exitcode=66")
    exe = str(tmp_path / "ring_tsan")
    r = subprocess.run(base + ["-o", exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-4000:]
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120, env=env)
    if "FATAL: ThreadSanitizer" in r.stdout and "data race" not in r.stdout:
        pytest.skip("ThreadSanitizer cannot run in this environment: %s" % r.stdout[-300:])
    assert r.returncode == 0 and "ring tsan test ok" in r.stdout and "data race" not in r.stdout, r.stdout[-4000:]
    bad = str(tmp_path / "ring_tsan_bad")
    r = subprocess.run(base + ["-DRING_TEST_SKIP_HOST_WAIT", "-o", bad], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-4000:]
    r = subprocess.run([bad], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120, env=env)
    assert r.returncode != 0 and ("data race" in r.stdout or "damaged" in r.stdout), r.stdout[-4000:]


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_no_mfma_result_is_read_too_early_in_the_shipped_kernels():
    """Static check of the gfx950 assembly (bench/check_mfma_hazards.py): no MFMA result is read or
    overwritten by a non-MFMA instruction fewer wait states later than the matrix pipe needs, along
    fall-through AND branch paths.  hipcc's own hazard cover stopped at a branch behind the last MFMA
    of conv_c64's row step once (wrong sums on the GPU, nothing at compile time, docs/KERNELS.md)."""
    r = subprocess.run(["python", os.path.join(ROOT, "bench", "check_mfma_hazards.py")], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-4000:]
    assert "conv_c64.hip" in r.stdout and "igemm.hip" in r.stdout
    import re
    counts = dict(re.findall(r"^(\S+\.hip)\s+(\d+) MFMAs", r.stdout, flags=re.M))
    assert int(counts["igemm.hip"]) > 1000 and int(counts["conv_c64.hip"]) > 300, counts     # the parser really saw the kernels
    assert " 0 of " in r.stdout and "kernels with scratch" in r.stdout


def test_mfma_hazard_checker_flags_a_known_bad_sequence(tmp_path):
    """The checker itself: a VALU read 2 wait states behind an MFMA across a branch is a finding, the
    same read behind enough s_nops is not, an accumulation chain on the same registers is not, and a
    kernel with scratch memory is reported (so a silent change of hipcc's output format cannot turn
    the tree-wide check into a no-op: that one also asserts non-zero MFMA counts)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_mfma_hazards", os.path.join(ROOT, "bench", "check_mfma_hazards.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    bad = """
_Z3badv:
\tv_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], v[12:15], v[0:3]
\tv_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], v[12:15], v[0:3]
\ts_cbranch_vccnz .LBB0_2
\tv_mov_b32_e32 v20, 0
.LBB0_2:
\ts_nop 0
\tv_cvt_pk_bf16_f32 v4, v0, v1
\ts_endpgm
_Z4goodv:
\tv_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], v[12:15], v[0:3]
\ts_cbranch_vccnz .LBB1_2
\tv_mov_b32_e32 v20, 0
.LBB1_2:
\ts_nop 7
\tv_cvt_pk_bf16_f32 v4, v0, v1
\ts_endpgm
"""
    f = tmp_path / "k.s"
    f.write_text(bad)
    kernels = chk.parse(str(f))
    assert set(kernels) == {"_Z3badv", "_Z4goodv"}
    found_bad = chk.check_kernel("_Z3badv", *kernels["_Z3badv"])
    found_good = chk.check_kernel("_Z4goodv", *kernels["_Z4goodv"])
    assert found_good == []
    assert len(found_bad) >= 1 and all(fd[0] == "_Z3badv" and "v_cvt_pk_bf16_f32" in fd[4] for fd in found_bad)
    assert min(fd[5] for fd in found_bad) <= 3 and found_bad[0][6] == 7       # wait states seen / needed
    assert chk.need_of("v_mfma_f32_32x32x16_bf16") > chk.need_of("v_mfma_f32_16x16x32_bf16")
