#!/usr/bin/env python3
"""sha256 of the gfx950 machine code of selected kernels inside libses_hip.so.

bench.py attaches profile-derived numbers (HBM traffic from PMC passes, VALU instruction counts from SQ passes) to
its JSON line only while the kernel they were collected on is still the kernel that runs; the profile files carry
the hash printed here (tools/collect_pmc.py, tools/prof_sq.sh).

    python tools/kernel_hash.py [lib.so] k_env_step_cartpole_v4 [more name fragments ...]

Pure Python: ELF64 section / symbol tables and the clang offload bundle inside .hip_fatbin are parsed directly.
"""
import hashlib
import os
import struct
import sys
import zlib

BUNDLE_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _sections(elf):
    assert elf[:4] == b"\x7fELF" and elf[4] == 2, "not an ELF64 file"
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
    secs = []
    for i in range(shnum):
        name, typ, _flags, addr, off, size, link, _info, _align, entsize = struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize)
        secs.append({"name_off": name, "type": typ, "addr": addr, "off": off, "size": size, "link": link, "entsize": entsize})
    strtab = secs[shstrndx]
    for s in secs:
        end = elf.index(b"\0", strtab["off"] + s["name_off"])
        s["name"] = elf[strtab["off"] + s["name_off"]:end].decode()
    return secs


def _code_objects(lib_bytes, arch="gfx950"):
    """Every gfx950 code object of the library: one clang offload bundle per translation unit (csrc/build.sh compiles the
    units separately, the linker concatenates their .hip_fatbin contributions)."""
    secs = _sections(lib_bytes)
    fat = next(s for s in secs if s["name"] == ".hip_fatbin")
    blob = lib_bytes[fat["off"]:fat["off"] + fat["size"]]
    if blob.find(BUNDLE_MAGIC) < 0 and blob[:4] == b"CCOB":   # compressed bundle: header, then one zlib / zstd stream
        raise RuntimeError("compressed offload bundle: rebuild with --no-offload-compress or hash with llvm tools")
    out, pos = [], blob.find(BUNDLE_MAGIC)
    assert pos >= 0, "no clang offload bundle in .hip_fatbin"
    while pos >= 0:
        nxt = blob.find(BUNDLE_MAGIC, pos + len(BUNDLE_MAGIC))
        b = blob[pos:nxt if nxt >= 0 else len(blob)]
        n, = struct.unpack_from("<Q", b, len(BUNDLE_MAGIC))
        p = len(BUNDLE_MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", b, p)
            triple = b[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if arch in triple and size:
                out.append(b[off:off + size])
        pos = nxt
    if not out:
        raise RuntimeError(f"no {arch} code object in the bundle")
    return out


def _code_object(lib_bytes, arch="gfx950"):
    """the first gfx950 code object (kept for callers that disassemble one unit); see _code_objects"""
    return _code_objects(lib_bytes, arch)[0]


def kernel_symbols(lib_path, fragment):
    """{mangled name: machine code bytes} of the FUNC symbols of the gfx950 code objects whose name contains fragment"""
    out = {}
    for co in _code_objects(open(lib_path, "rb").read()):
        secs = _sections(co)
        symtab = next(s for s in secs if s["name"] == ".symtab")
        strtab = secs[symtab["link"]]
        for i in range(symtab["size"] // 24):
            name_off, info, _other, shndx, value, size = struct.unpack_from("<IBBHQQ", co, symtab["off"] + 24 * i)
            if (info & 0xF) != 2 or size == 0 or shndx == 0 or shndx >= len(secs):       # STT_FUNC only
                continue
            end = co.index(b"\0", strtab["off"] + name_off)
            name = co[strtab["off"] + name_off:end].decode()
            if fragment in name:
                sec = secs[shndx]
                start = sec["off"] + (value - sec["addr"])
                out[name] = co[start:start + size]
    return out


def hash_kernels(lib_path, fragment):
    syms = kernel_symbols(lib_path, fragment)
    if not syms:
        raise RuntimeError(f"no kernel matching {fragment!r} in {lib_path}")
    h = hashlib.sha256()
    for name in sorted(syms):
        h.update(name.encode() + b"\0" + syms[name])
    return h.hexdigest()


if __name__ == "__main__":
    argv = sys.argv[1:]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = argv.pop(0) if argv and argv[0].endswith(".so") else os.path.join(root, "simple-es_amd", "libses_hip.so")
    for frag in argv or ["k_env_step_cartpole_v4", "k_rollout_cartpole_mlp"]:
        syms = kernel_symbols(lib, frag)
        print(frag, hash_kernels(lib, frag), f"({len(syms)} symbols, {sum(map(len, syms.values()))} bytes)")
