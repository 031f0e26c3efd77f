"""AddressSanitizer + UndefinedBehaviorSanitizer over the product's host-side parsers (image decoders, ONNX reader, planner) on the CPU
build — GPU sanitizers are not available on the pool.  tests/native/host_sanitize.cpp decodes every golden image and model, then ~9 000
damaged variants of them (truncations, byte flips, splats): each must come back as an error or a result, never as a sanitizer report.
Round 5 found and fixed two defects this way: a JPEG DC category taken unchecked from a damaged Huffman table (shift by > 31 bits in the
bit reader) and zero-length memcpy calls on null vector storage in the ONNX tensor reader."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "facerecognizeonnx_amd", "csrc")


@pytest.mark.timeout(600)
def test_host_parsers_survive_damaged_inputs_under_asan_ubsan(tmp_path):
    if not shutil.which("g++"):
        pytest.skip("no host compiler")
    exe = str(tmp_path / "host_sanitize")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined",
           "-o", exe, os.path.join(ROOT, "tests", "native", "host_sanitize.cpp"), os.path.join(CSRC, "image_io.cpp"),
           os.path.join(CSRC, "onnx_reader.cpp"), os.path.join(CSRC, "plan.cpp"), "-lz"]
    b = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=400)
    assert b.returncode == 0, b.stdout[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="allocator_may_return_null=1:max_allocation_size_mb=4096:detect_leaks=1")
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden"), str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=400)
    assert r.returncode == 0 and "0 failures" in r.stdout, r.stdout[-4000:]
