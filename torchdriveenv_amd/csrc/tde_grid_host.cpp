// tde_grid_host.cpp — the library's HOST grid builder (tde_gridbuild.h: tde_grid_build / tde_grid_free of include/tde_hip.h) as a
// translation unit of its own for the HOST compiler, so that this product code can be built and run under AddressSanitizer /
// UndefinedBehaviorSanitizer / ThreadSanitizer (`make -C torchdriveenv_amd/csrc san` -> torchdriveenv_amd/_san/): in libtde_hip.so it
// is compiled by hipcc inside tde_api.hip, where no host sanitizer reaches it.  Same source, same entry points, plus tde_last_error
// for the message; world.py loads such a build instead of libtde_hip.so when TDE_GRID_LIB names it (tests/test_sanitizers.py).
// With -DTDE_GRID_DRIVER it is a stand-alone program (no Python: ThreadSanitizer and a Python process with torch in it do not get
// along) that builds the index of a synthetic road network on 1, 3 and 8 threads and checks that the tables do not depend on the
// thread count.
#include <cstdio>
#include <cstring>

#include "../../include/tde_hip.h"

static thread_local char g_err[256] = "";
static int bad(const char *msg)
{
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return 1;
}
extern "C" __attribute__((visibility("default"))) const char *tde_last_error(void) { return g_err; }
extern "C" __attribute__((visibility("default"))) int tde_abi_version(void) { return TDE_ABI_VERSION; }

#include "tde_gridbuild.h"

#ifdef TDE_GRID_DRIVER
#include <cmath>
#include <vector>

// a ring road with spokes: ribbons of two triangles per metre-and-a-half, like synth.Town's streets
static void ribbon(std::vector<float> &tri, double x0, double y0, double x1, double y1, double width, double ds)
{
    const double L = std::hypot(x1 - x0, y1 - y0), tx = (x1 - x0) / L, ty = (y1 - y0) / L, nx = -ty * 0.5 * width, ny = tx * 0.5 * width;
    const int n = (int)std::ceil(L / ds);
    for (int i = 0; i < n; ++i) {
        const double a = L * i / n, b = L * (i + 1) / n;
        const double ax = x0 + tx * a, ay = y0 + ty * a, bx = x0 + tx * b, by = y0 + ty * b;
        const float q[12] = {(float)(ax - nx), (float)(ay - ny), (float)(bx - nx), (float)(by - ny), (float)(bx + nx), (float)(by + ny),
                             (float)(ax - nx), (float)(ay - ny), (float)(bx + nx), (float)(by + ny), (float)(ax + nx), (float)(ay + ny)};
        tri.insert(tri.end(), q, q + 12);
    }
}

int main()
{
    std::vector<float> tri;
    const int spokes = 7;
    for (int k = 0; k < 48; ++k) {
        const double a0 = 2 * M_PI * k / 48, a1 = 2 * M_PI * (k + 1) / 48;
        ribbon(tri, 120 * std::cos(a0), 120 * std::sin(a0), 120 * std::cos(a1), 120 * std::sin(a1), 7.0, 1.5);
    }
    for (int k = 0; k < spokes; ++k) {
        const double a = 2 * M_PI * k / spokes + 0.3;
        ribbon(tri, 0, 0, 150 * std::cos(a), 150 * std::sin(a), 7.0, 1.5);
    }
    const int n_tri = (int)(tri.size() / 6);
    tde_grid *ref = nullptr;
    if (tde_grid_build(tri.data(), n_tri, 0.5f, 0.25f, 0.05f, 2.0f, 1, &ref)) { fprintf(stderr, "build failed: %s\n", tde_last_error()); return 1; }
    const size_t ncell = (size_t)ref->nx * ref->ny, ntile = (size_t)(ref->nx / 4) * (ref->ny / 4);
    int rc = 0;
    for (int nt : {3, 8, 8}) {
        tde_grid *g = nullptr;
        if (tde_grid_build(tri.data(), n_tri, 0.5f, 0.25f, 0.05f, 2.0f, nt, &g)) { fprintf(stderr, "build failed: %s\n", tde_last_error()); return 1; }
        const bool same = g->nx == ref->nx && g->ny == ref->ny && g->n_records == ref->n_records && g->n_near_lists == ref->n_near_lists &&
                          !memcmp(g->cell_class, ref->cell_class, ncell) && !memcmp(g->cell_count, ref->cell_count, ncell) &&
                          !memcmp(g->cell_first, ref->cell_first, 4 * ncell) && !memcmp(g->cell_sub, ref->cell_sub, 4 * ncell) &&
                          !memcmp(g->rec_tri, ref->rec_tri, 4 * (size_t)ref->n_records) && !memcmp(g->rec_len, ref->rec_len, 4 * (size_t)ref->n_records) &&
                          !memcmp(g->tile_near, ref->tile_near, 4 * ntile);
        if (!same) { fprintf(stderr, "tables differ between 1 and %d threads\n", nt); rc = 1; }
        tde_grid_free(g);
    }
    // the refusals: messages, no crash, nothing leaked
    tde_grid *g = nullptr;
    const float nanv[6] = {NAN, 0, 1, 0, 0, 1};
    if (!tde_grid_build(nanv, 1, 0.5f, 0.25f, 0.05f, 2.0f, 2, &g) || !strstr(tde_last_error(), "non-finite")) rc = 1;
    const float huge[6] = {0, 0, 1e6f, 0, 0, 1e6f};
    if (!tde_grid_build(huge, 1, 0.5f, 0.25f, 0.05f, 2.0f, 2, &g) || !strstr(tde_last_error(), "larger cell")) rc = 1;
    printf("grid driver: %d triangles, %d x %d cells, %lld records, %lld near lists: %s\n", n_tri, ref->nx, ref->ny, (long long)ref->n_records,
           (long long)ref->n_near_lists, rc ? "FAILED" : "ok");
    tde_grid_free(ref);
    return rc;
}
#endif
