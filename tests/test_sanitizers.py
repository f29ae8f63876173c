"""The CPU-side code under AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers are not available on the
pool): hostlogic.cpp + spec.cpp of the library and the oracle, driven through their edge cases by tests/cpp/sanitize_cpu.cpp."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_logic_and_oracle_are_clean_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "sanitize_cpu")
    src = [os.path.join(ROOT, "tests", "cpp", "sanitize_cpu.cpp"), os.path.join(ROOT, "annonet_amd", "csrc", "hostlogic.cpp"),
           os.path.join(ROOT, "annonet_amd", "csrc", "spec.cpp"), os.path.join(ROOT, "oracle", "annonet_oracle.cpp")]
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fopenmp", "-mavx2", "-mfma", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", *src, "-o", exe])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="4")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr[-4000:]
    assert "sanitize_cpu ok" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
