"""Sanitizer runs of the host-side native code (CPU builds; GPU AddressSanitizer is not available on the pool): the readers on
good and damaged files, the post-processing on random motif families.  Skipped where g++ has no sanitizer runtime."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _have_asan(tmp_path):
    if not shutil.which("g++"):
        return False
    src = tmp_path / "t.cpp"
    src.write_text("int main() { return 0; }\n")
    return subprocess.run(["g++", "-fsanitize=address,undefined", str(src), "-o", str(tmp_path / "t")], capture_output=True).returncode == 0


@pytest.mark.timeout(900)
def test_host_readers_on_damaged_files_under_asan_and_ubsan(tmp_path):
    """tools/asan_reader: nmbed.cpp (text, gzip, BGZF + tabix with CRC-32 checks, FASTA) over ~100 good and damaged files — every
    file loads or is refused with a message, no sanitizer report."""
    if not _have_asan(tmp_path):
        pytest.skip("g++ -fsanitize=address,undefined is not usable here")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asan_reader", "run.py")], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "exit code 0" in r.stdout and "refused" in r.stdout


@pytest.mark.timeout(900)
def test_native_postprocessing_under_asan_and_ubsan(tmp_path):
    """tools/asan_post: nmpost.cpp on 400 random motif families with a hashing scorer, run twice, exports identical."""
    if not _have_asan(tmp_path):
        pytest.skip("g++ -fsanitize=address,undefined is not usable here")
    exe = str(tmp_path / "asan_post")
    c = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-pthread",
                        os.path.join(ROOT, "tools", "asan_post", "driver.cpp"), os.path.join(ROOT, "nanomotif_amd", "csrc", "nmpost.cpp"),
                        os.path.join(ROOT, "nanomotif_amd", "csrc", "nmsearch.cpp"), "-o", exe], capture_output=True, text=True)
    assert c.returncode == 0, c.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "identical exports" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.timeout(900)
def test_native_postprocessing_on_threads_under_tsan(tmp_path):
    """The same driver under ThreadSanitizer: run_post spreads its 400 tasks over threads (contiguous ranges, request lists joined in
    task order) — no race report, exports identical between the two runs."""
    src = tmp_path / "t.cpp"
    src.write_text("int main() { return 0; }\n")
    if not shutil.which("g++") or subprocess.run(["g++", "-fsanitize=thread", str(src), "-o", str(tmp_path / "t")], capture_output=True).returncode != 0:
        pytest.skip("g++ -fsanitize=thread is not usable here")
    exe = str(tmp_path / "tsan_post")
    c = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread",
                        os.path.join(ROOT, "tools", "asan_post", "driver.cpp"), os.path.join(ROOT, "nanomotif_amd", "csrc", "nmpost.cpp"),
                        os.path.join(ROOT, "nanomotif_amd", "csrc", "nmsearch.cpp"), "-o", exe], capture_output=True, text=True)
    assert c.returncode == 0, c.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, NM_POST_THREADS="6"))
    assert r.returncode == 0 and "identical exports" in r.stdout and "ThreadSanitizer" not in r.stderr, r.stdout[-2000:] + r.stderr[-3000:]
