"""Regenerates the golden vectors that come from code OUTSIDE this repository (container only;
needs /root/reference and /opt/rocm):

  tests/golden/obj_<name>.npz   (the 7 bundled meshes + tests/golden/quirks_{a,b}.obj) what the REFERENCE's own OBJ loader (objcore, compiled from
                                /root/reference by oracle/ref_build/Makefile, no stand-ins) holds
                                after loading each bundled mesh: points, normals, faces,
                                facenormals (ragged lists flattened with offsets), bbox
  tests/golden/thrust_rng.npz   raw values and uniforms of the real thrust::default_random_engine /
                                uniform_real_distribution<float>(0,1) (the image's rocThrust) for 64
                                seeds x 4096 draws

  tests/golden/ref_helpers.npz  outputs of the REFERENCE's own text of the per-path helper functions
                                (path_tracer_kernel.cu:46-57, 75-84, 194-475, compiled unmodified by
                                oracle/ref_build/Makefile against the CUDA headers of the image's Triton
                                wheel) for 102 400 inputs per function; the inputs are recomputed by
                                tests/ref_helper_inputs.py, the file keeps their SHA-256

Only outputs are stored: no source text of the reference enters the repository."""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_helper_inputs  # noqa: E402
REF = "/root/reference/GPUPathTracer"
GOLD = os.path.join(ROOT, "tests", "golden")
BIN = os.path.join(ROOT, "oracle", "_ref")
MESHES = ("box", "box2", "cone", "diamond", "coil", "monkey", "cow")
M = 2147483647


def randhash(a):
    a &= 0xFFFFFFFF
    a = ((a + 0x7ed55d16) + (a << 12)) & 0xFFFFFFFF
    a = ((a ^ 0xc761c23c) ^ (a >> 19)) & 0xFFFFFFFF
    a = ((a + 0x165667b1) + (a << 5)) & 0xFFFFFFFF
    a = ((a + 0xd3a2646c) ^ (a << 9)) & 0xFFFFFFFF
    a = ((a + 0xfd7046c5) + (a << 3)) & 0xFFFFFFFF
    a = ((a ^ 0xb55a4f09) ^ (a >> 16)) & 0xFFFFFFFF
    return a


def rng_seeds():
    """64 seeds: the edge cases of seeding (0, m, multiples of m, 2^32-1), states whose next value maps
    to u == 1.0f, and seeds formed the way the kernels form them (products of three hashes)."""
    inv = pow(48271, -1, M)
    seeds = [0, 1, 2, M - 1, M, M + 1, M + 5, 2 * M, 2 * M + 1, 0xFFFFFFFF, 12345, 399268537]
    seeds += [((M - 1 - k) * inv) % M for k in (0, 1, 31, 62, 63, 64)]     # successor x with x - 1 >= 2^31 - 64 (or just below)
    seeds += [inv]                                                          # successor x = 1 -> u = 0
    k = 0
    while len(seeds) < 64:
        c, p, d = k % 5, (k * 7919 + 3) % 2073600, k % 8
        seeds.append((randhash(c) * randhash(p) * randhash(d)) & 0xFFFFFFFF)                 # trace kernel (:492)
        if len(seeds) < 64:
            seeds.append((randhash(c) * randhash(p) * randhash(c)) & 0xFFFFFFFF)             # camera kernel (:139)
        k += 1
    return seeds[:64]


def read_ints(buf, pos, n):
    return np.frombuffer(buf, dtype="<i4", count=n, offset=pos), pos + 4 * n


def main():
    if not os.path.isdir(REF):
        sys.exit("the reference checkout is not present: goldens can only be regenerated in the build container")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle", "ref_build")])
    os.makedirs(GOLD, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        # the reference's bundled meshes, and two small files of this repository (tests/golden/quirks_*.obj)
        # that exercise the loader's syntaxes: v//n, v/t/n, bare v, polygons, comments, groups
        for name in MESHES + ("quirks_a", "quirks_b"):
            out = os.path.join(tmp, name + ".bin")
            src = os.path.join(GOLD, name + ".obj") if name.startswith("quirks") else os.path.join(REF, name + ".obj")
            subprocess.check_call([os.path.join(BIN, "objcore_dump"), src, out],
                                  stdout=subprocess.DEVNULL)
            buf = open(out, "rb").read()
            (npts, nnrm, nfaces, nfn), pos = read_ints(buf, 0, 4)
            points = np.frombuffer(buf, "<f4", npts * 3, pos).reshape(npts, 3).copy(); pos += 12 * npts
            normals = np.frombuffer(buf, "<f4", nnrm * 3, pos).reshape(nnrm, 3).copy(); pos += 12 * nnrm
            lists = []
            for count in (nfaces, nfn):
                offs, flat = [0], []
                for _ in range(count):
                    (k,), pos = read_ints(buf, pos, 1)
                    idx, pos = read_ints(buf, pos, int(k))
                    flat.append(idx)
                    offs.append(offs[-1] + int(k))
                lists.append((np.array(offs, np.int32), np.concatenate(flat).astype(np.int32) if flat else np.zeros(0, np.int32)))
            bbox = np.frombuffer(buf, "<f4", 6, pos).copy(); pos += 24
            assert pos == len(buf)
            np.savez_compressed(os.path.join(GOLD, "obj_%s.npz" % name), points=points, normals=normals,
                                face_offsets=lists[0][0], face_indices=lists[0][1],
                                facenormal_offsets=lists[1][0], facenormal_indices=lists[1][1], bbox=bbox)
            print("obj_%s: %d points, %d normals, %d faces" % (name, npts, nnrm, nfaces))
        seeds = rng_seeds()
        n_draws = 4096
        out = os.path.join(tmp, "rng.bin")
        subprocess.run([os.path.join(BIN, "thrust_rng_dump"), str(n_draws), out], input="\n".join(map(str, seeds)) + "\n",
                       text=True, check=True)
        buf = open(out, "rb").read()
        n, nn = struct.unpack_from("<ii", buf, 0)
        assert (n, nn) == (len(seeds), n_draws)
        pos = 8
        got_seeds, raw, uni = [], [], []
        for _ in range(n):
            got_seeds.append(struct.unpack_from("<I", buf, pos)[0]); pos += 4
            raw.append(np.frombuffer(buf, "<u4", nn, pos)); pos += 4 * nn
            uni.append(np.frombuffer(buf, "<f4", nn, pos)); pos += 4 * nn
        kat = struct.unpack_from("<I", buf, pos)[0]
        assert got_seeds == seeds and kat == 399268537
        # raw values in full for the first 64 draws (the uniforms are a function of them); uniforms for all draws
        np.savez_compressed(os.path.join(GOLD, "thrust_rng.npz"), seeds=np.array(seeds, np.uint32),
                            raw=np.stack(raw)[:, :64].copy(), uniform=np.stack(uni), minstd_10000th=np.uint32(kat))
        print("thrust_rng: %d seeds x %d draws; u == 1.0 occurs %d times, u == 0 %d times"
              % (n, nn, int((np.stack(uni) == 1.0).sum()), int((np.stack(uni) == 0.0).sum())))
        helper_goldens(tmp)


def helper_goldens(tmp):
    exe = os.path.join(BIN, "helpers_dump")
    if not os.path.isfile(exe):
        print("ref_helpers: skipped (oracle/_ref/helpers_dump was not built: no CUDA headers in this image)")
        return
    arrays = {}
    for name in ref_helper_inputs.FUNCTIONS:
        n_in, n_out = ref_helper_inputs.LAYOUT[name]
        words = ref_helper_inputs.make(name)
        fin, fout = os.path.join(tmp, name + ".in"), os.path.join(tmp, name + ".out")
        words.tofile(fin)
        subprocess.check_call([exe, name, str(len(words)), fin, fout])
        out = np.fromfile(fout, dtype="<u4").reshape(len(words), n_out)
        arrays["out_" + name] = out
        arrays["sha_" + name] = np.array(ref_helper_inputs.digest(words))
        print("ref_helpers: %-12s %d records, %d words in / %d out" % (name, len(words), n_in, n_out))
    np.savez_compressed(os.path.join(GOLD, "ref_helpers.npz"), **arrays)


if __name__ == "__main__":
    main()
