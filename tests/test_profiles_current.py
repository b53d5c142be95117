"""The counter profiles of the CURRENT round must describe the library that ships.

bench.py attaches profile-derived figures (HBM traffic from the PMC passes, VALU / MFMA counts from the SQ passes) to
its JSON line only while the profiled kernel's machine code is the code that runs; a kernel edit after the last profile
pass silently drops them from the line of record.  This test makes that visible on the CPU: every profiles/rNN_*.json
of the newest round that carries a `kernel_code_sha256` and the `kernel_match` fragment it was taken over must equal
tools/kernel_hash.py on the in-tree libses_hip.so.  (Older rounds' files are history and are not checked.)"""
import glob
import json
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_hash  # noqa: E402

LIB = os.path.join(ROOT, "simple-es_amd", "libses_hip.so")


def newest_round():
    tags = {m.group(1) for f in glob.glob(os.path.join(ROOT, "profiles", "r*_*.json"))
            for m in [re.match(r"(r\d+)_", os.path.basename(f))] if m}
    return max(tags, key=lambda t: int(t[1:])) if tags else None


def hashed_profiles(tag):
    out = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", tag + "_*.json"))):
        try:
            d = json.load(open(f))
        except ValueError:
            continue
        if isinstance(d, dict) and d.get("kernel_code_sha256") and d.get("kernel_match"):
            out.append((os.path.basename(f), d["kernel_match"], d["kernel_code_sha256"]))
    return out


def test_kernel_hash_reads_the_in_tree_library():
    assert os.path.exists(LIB), "build the library first (python __graft_entry__.py)"
    h = kernel_hash.hash_kernels(LIB, "k_env_step_cartpole_v4")
    assert re.fullmatch(r"[0-9a-f]{64}", h)


def test_profiles_of_the_newest_round_match_the_shipped_kernels():
    tag = newest_round()
    if tag is None:
        pytest.skip("no profiles committed")
    stale = []
    for name, match, want in hashed_profiles(tag):
        got = kernel_hash.hash_kernels(LIB, match)
        if got != want:
            stale.append(f"{name}: profiled {match} = {want[:12]}..., library has {got[:12]}...")
    assert not stale, ("profiles collected on different machine code than the in-tree library -- re-run "
                       "tools/gpu_profile_round.sh and the collect_* summarisers as the last step:\n  " + "\n  ".join(stale))


def test_stream_probe_moves_the_env_step_kernels_thirteen_streams():
    """ses_stream_probe is the ceiling bench.py prints next to the env-step kernel: its machine code must hold exactly the
    env-step kernel's 7 x 16-byte non-temporal loads and 6 x 16-byte non-temporal stores (a store-back of an unchanged value
    is something the optimiser drops unless told otherwise -- it did once)."""
    import shutil
    import subprocess
    import tempfile
    objdump = shutil.which("llvm-objdump") or "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    text = ""
    for co in kernel_hash._code_objects(open(LIB, "rb").read()):
        if b"k_stream_probe13" not in co:
            continue
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            text += subprocess.run([objdump, "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True, timeout=300).stdout
    counts = {}
    for frag in ("k_stream_probe13", "k_env_step_cartpole_v4ILb1"):
        body = re.search(r"<[^>]*" + frag + r"[^>]*>:(.*?)s_endpgm", text, flags=re.S)
        assert body, frag
        counts[frag] = (len(re.findall(r"global_load_dwordx4 .* nt", body.group(1))),
                        len(re.findall(r"global_store_dwordx4 .* nt", body.group(1))))
    assert counts["k_stream_probe13"] == (7, 6) == counts["k_env_step_cartpole_v4ILb1"], counts
