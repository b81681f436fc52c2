"""Host replay of the tile GEMM's index arithmetic (csrc/gemm_tile_index.hpp): DMA image, fragment reads, LDS bank
slots, accumulator layout and the SwiGLU row pairing, compiled with g++ - runs without a GPU."""
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gemm_tile_index_replay(tmp_path):
    exe = tmp_path / "gemm_tile_index_check"
    subprocess.run(["g++", "-O1", "-std=c++17", "-I", os.path.join(REPO, "nano-vllm-ascend_amd", "csrc"),
                    os.path.join(REPO, "tests", "host", "gemm_tile_index_check.cpp"), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ok" in out.stdout
