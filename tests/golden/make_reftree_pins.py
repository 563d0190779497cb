"""Golden vectors from code the reference tree itself holds (the only executable
pins available, since EMAN2 is not vendored).  Run in the BUILD container, where /root/reference is
mounted:

    python tests/golden/make_reftree_pins.py

It never copies reference text into the repository: the text is read at run time, compiled as host
C++ in a temporary directory, executed on seeded inputs, and only the numeric inputs / outputs are
written as fixtures:

  rot_shift2d_ref.npz  rot_scale_trans2D_background + quadri_background exactly as
                       notebook/02_CuPy_Image_Processing_rot_shift2d.ipynb cell 2 spells them (the reference's
                       own restatement of what EMAN2's rot_shift2D does), on random images / parameters.
                       Angles are restricted to those where cosf/sinf and the double-precision cos/sin rounded
                       to float agree on this machine, so the vectors do not depend on which overload the
                       compiler picks for `cos(float)`.
  tanl_ref.npz         the tangent low-pass profile H(d) of cuda/gpu_aln_noref.cu:786-816
                       (cu_apply_tanl_filter_to_tex), evaluated by running that kernel text on an all-ones
                       spectrum, together with the frequency d the kernel assigns to every cell.
  filter_log.json      the "Tangent filter: cut-off frequency / fall-off" lines of the run log in
                       notebook/00_Multireference_Alignment.ipynb cell 3 (clamp evidence: 0.120 / 0.200).
"""
import json
import math
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

GLUE = r"""
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
struct idx3 { unsigned x, y, z; };
static idx3 blockIdx, blockDim, threadIdx;
#define __device__
#define __global__
#define __restrict__
"""


def notebook_kernel_text():
    nb = json.load(open(os.path.join(REF, "notebook", "02_CuPy_Image_Processing_rot_shift2d.ipynb")))
    src = "".join(nb["cells"][2]["source"])
    m = re.search(r"cp\.RawKernel\(r'''(.*?)'''", src, re.S)
    assert m, "RawKernel text not found"
    text = m.group(1)
    text = text.replace('extern "C" __global__', "")       # host function
    return text


def build_and_run(code, args, workdir):
    src = os.path.join(workdir, "pin.cpp")
    exe = os.path.join(workdir, "pin")
    with open(src, "w") as f:
        f.write(code)
    subprocess.check_call(["g++", "-O0", "-ffp-contract=off", "-o", exe, src, "-lm"])
    subprocess.check_call([exe] + [str(a) for a in args])


def make_rot_shift(workdir):
    rng = np.random.Generator(np.random.PCG64(4242))
    cases = []
    for nx in (90, 33, 64):
        c = 8
        imgs = rng.standard_normal((c, nx, nx)).astype(np.float32)
        ang = rng.uniform(0, 360, 4 * c).astype(np.float32)
        # keep angles for which the float and the rounded double evaluation of cos / sin agree
        rad = (ang * np.float32(3.14159265358979323846) / np.float32(180.0)).astype(np.float32)
        ok = [np.float32(math.cos(float(a))) == np.cos(a, dtype=np.float32) and
              np.float32(math.sin(float(a))) == np.sin(a, dtype=np.float32) for a in rad]
        ang = ang[np.array(ok)][:c]
        assert len(ang) == c
        ang[0] = 0.0
        dx = rng.uniform(-6, 6, c).astype(np.float32)
        dy = rng.uniform(-6, 6, c).astype(np.float32)
        dx[1] = 3.0; dy[1] = -2.0            # integer shifts
        dx[2] = nx + 1.5                      # restrict2 wrap
        cases.append((nx, imgs, ang, dx, dy))
    harness = GLUE + notebook_kernel_text() + r"""
int main(int argc, char **argv)
{
    int nx = atoi(argv[1]), c = atoi(argv[2]);
    FILE *f = fopen(argv[3], "rb");
    size_t n = (size_t)nx * nx * c;
    float *src = (float *)malloc(n * 4), *dst = (float *)calloc(n, 4);
    float *ang = (float *)malloc(c * 4), *dx = (float *)malloc(c * 4), *dy = (float *)malloc(c * 4), *sc = (float *)malloc(c * 4);
    if (fread(src, 4, n, f) != n || fread(ang, 4, c, f) != (size_t)c || fread(dx, 4, c, f) != (size_t)c || fread(dy, 4, c, f) != (size_t)c) return 2;
    fclose(f);
    for (int i = 0; i < c; i++) sc[i] = 1.0f;
    blockDim.x = blockDim.y = blockDim.z = 1;
    for (int z = 0; z < c; z++)
        for (int iy = 0; iy < nx; iy++)
            for (int ix = 0; ix < nx; ix++) {
                blockIdx.x = ix; blockIdx.y = iy; blockIdx.z = z;
                threadIdx.x = threadIdx.y = threadIdx.z = 0;
                float a0 = ang[z], x0 = dx[z], y0 = dy[z];      /* the kernel rewrites delx / dely in place (restrict2) */
                rot_scale_trans2D_background(dst, src, nx, nx, c, ang, dx, dy, sc);
                ang[z] = a0; dx[z] = x0; dy[z] = y0;
            }
    f = fopen(argv[4], "wb");
    fwrite(dst, 4, n, f);
    fclose(f);
    return 0;
}
"""
    out = {}
    for k, (nx, imgs, ang, dx, dy) in enumerate(cases):
        fin, fout = os.path.join(workdir, "in%d.bin" % k), os.path.join(workdir, "out%d.bin" % k)
        with open(fin, "wb") as f:
            f.write(imgs.tobytes()); f.write(ang.tobytes()); f.write(dx.tobytes()); f.write(dy.tobytes())
        build_and_run(harness, [nx, imgs.shape[0], fin, fout], workdir)
        res = np.fromfile(fout, np.float32).reshape(imgs.shape)
        out["img%d" % k] = imgs; out["ang%d" % k] = ang; out["dx%d" % k] = dx; out["dy%d" % k] = dy
        out["out%d" % k] = res
    out["ncase"] = np.int32(len(cases))
    np.savez_compressed(os.path.join(HERE, "rot_shift2d_ref.npz"), **out)
    print("rot_shift2d_ref.npz:", {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.ndim})


def tanl_kernel_text():
    lines = open(os.path.join(REF, "cuda", "gpu_aln_noref.cu"), encoding="utf-8", errors="replace").read().split("\n")
    start = next(i for i, l in enumerate(lines) if "void cu_apply_tanl_filter_to_tex" in l)
    depth, end = 0, None
    for i in range(start, len(lines)):
        depth += lines[i].count("{") - lines[i].count("}")
        if depth == 0 and "}" in lines[i] and i > start:
            end = i
            break
    pi = next(l for l in open(os.path.join(REF, "cuda", "gpu_aln_common.h")) if l.startswith("#define PI"))
    return pi + "\n".join(lines[start:end + 1])


def make_tanl(workdir):
    harness = GLUE + "struct cuComplex { float x, y; };\n" + tanl_kernel_text() + r"""
int main(int argc, char **argv)
{
    int nx = atoi(argv[1]);
    float fl = atof(argv[2]), aa = atof(argv[3]);
    int nxh = nx / 2 + 1;
    cuComplex *img = (cuComplex *)malloc(sizeof(cuComplex) * nxh * nx);
    for (int i = 0; i < nxh * nx; i++) { img[i].x = 1.0f; img[i].y = 1.0f; }
    blockDim.x = nxh; blockIdx.x = 0;
    for (int t = 0; t < nxh; t++) {
        threadIdx.x = t;
        cu_apply_tanl_filter_to_tex(img, sizeof(cuComplex) * nxh, nx, fl, aa);
    }
    FILE *f = fopen(argv[4], "wb");
    fwrite(img, sizeof(cuComplex), (size_t)nxh * nx, f);
    fclose(f);
    return 0;
}
"""
    out = {}
    k = 0
    for nx in (90, 64):
        for fl, aa in ((0.12, 0.2), (0.25, 0.1), (0.4, 0.05)):
            fout = os.path.join(workdir, "tanl%d.bin" % k)
            build_and_run(harness, [nx, fl, aa, fout], workdir)
            h = np.fromfile(fout, np.float32).reshape(nx, nx // 2 + 1, 2)
            assert np.array_equal(h[..., 0], h[..., 1])
            # the frequency the kernel assigns to cell (ky, kx): x = kx / (nx/2+1) * 0.5 ; y = min(ky, nx-ky) / (nx/2) * 0.5
            nxh = nx // 2 + 1
            x = (np.arange(nxh, dtype=np.float32) / np.float32(nxh) * np.float32(0.5))[None, :]
            ky = np.arange(nx)
            y = np.where(ky < nx / 2.0, ky, nx - ky).astype(np.float32)[:, None] / np.float32(nx / 2.0) * np.float32(0.5)
            d = np.sqrt(x * x + y * y).astype(np.float32)
            out["H%d" % k] = h[..., 0].copy(); out["d%d" % k] = d
            out["par%d" % k] = np.array([nx, fl, aa], np.float64)
            k += 1
    out["ncase"] = np.int32(k)
    np.savez_compressed(os.path.join(HERE, "tanl_ref.npz"), **out)
    print("tanl_ref.npz: %d cases" % k)


def make_filter_log():
    nb = json.load(open(os.path.join(REF, "notebook", "00_Multireference_Alignment.ipynb")))
    txt = "".join(sum([o.get("text", []) for o in nb["cells"][3].get("outputs", [])], []))
    rows = []
    for l in txt.splitlines():
        m = re.search(r"cut-off frequency\s*=\s*([0-9.]+)\s+fall-off\s*=\s*([0-9.]+)", l)
        if m:
            r = [float(m.group(1)), float(m.group(2))]
            if r not in rows:
                rows.append(r)
    with open(os.path.join(HERE, "filter_log.json"), "w") as f:
        json.dump({"source": "notebook/00_Multireference_Alignment.ipynb cell 3 stdout", "cutoff_falloff": rows}, f, indent=1)
    print("filter_log.json: %d distinct lines, min cut-off %.3f, max fall-off %.3f" % (len(rows), min(r[0] for r in rows), max(r[1] for r in rows)))


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    with tempfile.TemporaryDirectory() as wd:
        make_rot_shift(wd)
        make_tanl(wd)
    make_filter_log()
