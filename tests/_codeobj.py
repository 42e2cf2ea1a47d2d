"""Per-kernel register / spill metadata of the built library, read out of the gfx950 code objects embedded in the .so (test infrastructure).

The .hip_fatbin section holds one clang offload bundle per translation unit; every bundle entry for an amdgcn target is an ELF code
object whose NT_AMDGPU_METADATA note is MessagePack with one map per kernel (.name, .vgpr_count, .agpr_count, .sgpr_count,
.vgpr_spill_count, .sgpr_spill_count, .private_segment_fixed_size, .group_segment_fixed_size ...).  No external tool is needed."""
import struct

import msgpack

BUNDLE_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _elf_sections(d):
    assert d[:4] == b"\x7fELF" and d[4] == 2, "not an ELF64 file"
    shoff, = struct.unpack_from("<Q", d, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", d, 0x3A)
    secs = []
    for i in range(shnum):
        name, typ, flags, addr, off, size = struct.unpack_from("<IIQQQQ", d, shoff + i * shentsize)
        secs.append((name, typ, off, size))
    stroff = secs[shstrndx][2]
    out = {}
    for name, typ, off, size in secs:
        end = d.index(b"\0", stroff + name)
        out.setdefault(d[stroff + name:end].decode(), []).append((typ, off, size))
    return out


def code_objects(so_path):
    """The amdgcn ELF images inside the library's fat binary: [(target triple, bytes)]."""
    d = open(so_path, "rb").read()
    typ, off, size = _elf_sections(d)[".hip_fatbin"][0]
    sec = d[off:off + size]
    out = []
    pos = sec.find(BUNDLE_MAGIC)
    while pos >= 0:
        n, = struct.unpack_from("<Q", sec, pos + len(BUNDLE_MAGIC))
        p = pos + len(BUNDLE_MAGIC) + 8
        for _ in range(n):
            eoff, esize, tsize = struct.unpack_from("<QQQ", sec, p)
            triple = sec[p + 24:p + 24 + tsize].decode()
            p += 24 + tsize
            if "amdgcn" in triple and esize:
                out.append((triple, sec[pos + eoff:pos + eoff + esize]))
        pos = sec.find(BUNDLE_MAGIC, pos + 1)
    return out


def _notes(elf):
    secs = _elf_sections(elf)
    for name, lst in secs.items():
        for typ, off, size in lst:
            if typ != 7:          # SHT_NOTE
                continue
            p = off
            while p + 12 <= off + size:
                namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
                p += 12
                nm = elf[p:p + namesz].rstrip(b"\0")
                p += (namesz + 3) & ~3
                desc = elf[p:p + descsz]
                p += (descsz + 3) & ~3
                yield nm, ntype, desc


def kernel_metadata(so_path):
    """{demangled-or-mangled kernel name: metadata map} over every gfx950 code object of the library."""
    out = {}
    for triple, elf in code_objects(so_path):
        for nm, ntype, desc in _notes(elf):
            if nm == b"AMDGPU" and ntype == 32:      # NT_AMDGPU_METADATA
                md = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                for k in md.get("amdhsa.kernels", []):
                    out[k[".name"]] = k
    return out


def summary(so_path):
    rows = []
    for name, k in sorted(kernel_metadata(so_path).items()):
        rows.append((name, k.get(".vgpr_count", 0), k.get(".agpr_count", 0), k.get(".sgpr_count", 0), k.get(".vgpr_spill_count", 0),
                     k.get(".sgpr_spill_count", 0), k.get(".private_segment_fixed_size", 0), k.get(".group_segment_fixed_size", 0)))
    return rows


if __name__ == "__main__":
    import sys
    for r in summary(sys.argv[1]):
        if len(sys.argv) < 3 or any(s in r[0] for s in sys.argv[2:]):
            print("%-110s vgpr %3d agpr %3d sgpr %3d  spill v %3d s %3d  scratch %4d  lds %6d" % ((r[0][:110],) + r[1:]))
