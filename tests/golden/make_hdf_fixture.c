/* Generates tests/golden/eman2_mdf_sample.hdf with the HDF5 C library: the EMAN2 "MDF" stack layout the
 * reference reads through EMAN2 (cuda/EMAN2_test.ipynb cell 4: f['MDF']['images'][str(i)]['image']):
 *   /MDF/images            attribute imageid_max (int)
 *   /MDF/images/<i>        attributes EMAN.nx, EMAN.ny, EMAN.nz (int), EMAN.apix_x (float), EMAN.ptcl_repr (int)
 *   /MDF/images/<i>/image  float32 [ny][nx], contiguous
 * Build + run here (the library ships with /opt/conda in the build container only):
 *   /opt/conda/bin/h5cc -o /tmp/mkfix tests/golden/make_hdf_fixture.c && /tmp/mkfix tests/golden/eman2_mdf_sample.hdf
 * The pixel values are value(i, y, x) = i * 1000 + y * 16 + x + 0.25, so the reader test needs no second file. */
#include <hdf5.h>
#include <stdio.h>
#include <stdlib.h>

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

int main(int argc, char **argv)
{
    const int n = 37, nx = 12, ny = 10;      /* 37 children: more than one symbol-table node (2 * 16 entries) */
    hid_t f = H5Fcreate(argc > 1 ? argv[1] : "eman2_mdf_sample.hdf", H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    hid_t mdf = H5Gcreate2(f, "/MDF", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    hid_t imgs = H5Gcreate2(f, "/MDF/images", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    put_int(imgs, "imageid_max", n - 1);
    float *buf = (float *)malloc(sizeof(float) * nx * ny);
    for (int i = 0; i < n; i++) {
        char name[64];
        snprintf(name, sizeof(name), "/MDF/images/%d", i);
        hid_t g = H5Gcreate2(f, name, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        put_int(g, "EMAN.nx", nx); put_int(g, "EMAN.ny", ny); put_int(g, "EMAN.nz", 1);
        put_float(g, "EMAN.apix_x", 1.25f); put_int(g, "EMAN.ptcl_repr", i % 3);
        for (int y = 0; y < ny; y++) for (int x = 0; x < nx; x++) buf[y * nx + x] = i * 1000 + y * 16 + x + 0.25f;
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
