/* Generates tests/golden/eman2_mdf_strings.hdf with the HDF5 C library: an EMAN2 "MDF" stack whose image headers hold
 * the kinds of items real stacks carry besides numbers -- a fixed-length string (EMAN.source_path), a variable-length
 * string (EMAN.ctf, which EMAN2 writes as a vlen string) and a double (EMAN.apix_y) -- so that the header write-back
 * (cryo_ralib_amd/mdfio.py: write_alignment_headers) can be tested against what it must carry over or refuse.
 *   /opt/conda/bin/h5cc -o /tmp/mkfix2 tests/golden/make_hdf_strings_fixture.c && /tmp/mkfix2 tests/golden/eman2_mdf_strings.hdf
 * Pixels: value(i, y, x) = i * 100 + y * 8 + x + 0.5. */
#include <hdf5.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void put_int(hid_t loc, const char *name, int v)
{
    hid_t sp = H5Screate(H5S_SCALAR), a = H5Acreate2(loc, name, H5T_NATIVE_INT, sp, H5P_DEFAULT, H5P_DEFAULT);
    H5Awrite(a, H5T_NATIVE_INT, &v); H5Aclose(a); H5Sclose(sp);
}
static void put_float(hid_t loc, const char *name, float v)
{
    hid_t sp = H5Screate(H5S_SCALAR), a = H5Acreate2(loc, name, H5T_NATIVE_FLOAT, sp, H5P_DEFAULT, H5P_DEFAULT);
    H5Awrite(a, H5T_NATIVE_FLOAT, &v); H5Aclose(a); H5Sclose(sp);
}
static void put_double(hid_t loc, const char *name, double v)
{
    hid_t sp = H5Screate(H5S_SCALAR), a = H5Acreate2(loc, name, H5T_NATIVE_DOUBLE, sp, H5P_DEFAULT, H5P_DEFAULT);
    H5Awrite(a, H5T_NATIVE_DOUBLE, &v); H5Aclose(a); H5Sclose(sp);
}
static void put_fixed_string(hid_t loc, const char *name, const char *v)
{
    hid_t t = H5Tcopy(H5T_C_S1);
    H5Tset_size(t, strlen(v) + 1);
    hid_t sp = H5Screate(H5S_SCALAR), a = H5Acreate2(loc, name, t, sp, H5P_DEFAULT, H5P_DEFAULT);
    H5Awrite(a, t, v); H5Aclose(a); H5Sclose(sp); H5Tclose(t);
}
static void put_vlen_string(hid_t loc, const char *name, const char *v)
{
    hid_t t = H5Tcopy(H5T_C_S1);
    H5Tset_size(t, H5T_VARIABLE);
    hid_t sp = H5Screate(H5S_SCALAR), a = H5Acreate2(loc, name, t, sp, H5P_DEFAULT, H5P_DEFAULT);
    H5Awrite(a, t, &v); H5Aclose(a); H5Sclose(sp); H5Tclose(t);
}

int main(int argc, char **argv)
{
    const int n = 5, nx = 8, ny = 6;
    hid_t f = H5Fcreate(argc > 1 ? argv[1] : "eman2_mdf_strings.hdf", H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    hid_t mdf = H5Gcreate2(f, "/MDF", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    hid_t imgs = H5Gcreate2(f, "/MDF/images", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    put_int(imgs, "imageid_max", n - 1);
    float *buf = (float *)malloc(sizeof(float) * nx * ny);
    for (int i = 0; i < n; i++) {
        char name[64], src[64];
        snprintf(name, sizeof(name), "/MDF/images/%d", i);
        snprintf(src, sizeof(src), "mics/micrograph_%02d.hdf", i);
        hid_t g = H5Gcreate2(f, name, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        put_int(g, "EMAN.nx", nx); put_int(g, "EMAN.ny", ny); put_int(g, "EMAN.nz", 1);
        put_float(g, "EMAN.apix_x", 1.25f); put_double(g, "EMAN.apix_y", 1.2500000001);
        put_int(g, "EMAN.source_n", 3 * i);
        put_fixed_string(g, "EMAN.source_path", src);
        put_vlen_string(g, "EMAN.ctf", "O1.8 200 2.0 0.1 0 1.25 0 0");
        for (int y = 0; y < ny; y++) for (int x = 0; x < nx; x++) buf[y * nx + x] = i * 100 + y * 8 + x + 0.5f;
        hsize_t dims[2] = {(hsize_t)ny, (hsize_t)nx};
        hid_t sp = H5Screate_simple(2, dims, NULL);
        hid_t d = H5Dcreate2(g, "image", H5T_NATIVE_FLOAT, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        H5Dwrite(d, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf);
        H5Dclose(d); H5Sclose(sp); H5Gclose(g);
    }
    free(buf);
    H5Gclose(imgs); H5Gclose(mdf); H5Fclose(f);
    return 0;
}
