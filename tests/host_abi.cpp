// A plain C++ host of the C ABI (no Python, no ctypes): what adapter/ORBextractor.cc and adapter/SurfelFusion.cpp do with
// cv::Mat / std::vector replaced by files.  tests/test_host_abi_gpu.py builds it with g++, runs it on the GPU box and compares
// its outputs with the CPU oracle.
//   host_abi <dir> <width> <height> <fx> <fy> <cx> <cy> <n_local> <ref>
// reads  <dir>/gray.bin (u8) depth.bin (f32) member.bin (i32, half resolution) pose.bin (16 f32, column-major) local.bin (msl_surfel)
// writes <dir>/kps.bin desc.bin local_out.bin new.bin tables.bin
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "msl.h"

template <typename T>
static std::vector<T> slurp(const std::string &path, size_t n) {
    std::vector<T> v(n);
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f || std::fread(v.data(), sizeof(T), n, f) != n) { std::fprintf(stderr, "cannot read %s\n", path.c_str()); std::exit(2); }
    std::fclose(f);
    return v;
}
template <typename T>
static void dump(const std::string &path, const T *p, size_t n) {
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f || std::fwrite(p, sizeof(T), n, f) != n) { std::fprintf(stderr, "cannot write %s\n", path.c_str()); std::exit(2); }
    std::fclose(f);
}
#define CHECK(call)                                                                          \
    do {                                                                                     \
        if ((call) != MSL_OK) { std::fprintf(stderr, "%s: %s\n", #call, msl_last_error()); return 1; } \
    } while (0)

int main(int argc, char **argv) {
    if (argc != 10) return 2;
    const std::string dir = argv[1];
    const int W = std::atoi(argv[2]), H = std::atoi(argv[3]);
    const float fx = (float)std::atof(argv[4]), fy = (float)std::atof(argv[5]), cx = (float)std::atof(argv[6]), cy = (float)std::atof(argv[7]);
    const size_t nLocal = (size_t)std::atoll(argv[8]);
    const int ref = std::atoi(argv[9]);
    if (msl_device_count() < 1) { std::fprintf(stderr, "no gfx950 device\n"); return 3; }
    const auto gray = slurp<uint8_t>(dir + "/gray.bin", (size_t)W * H);
    const auto depth = slurp<float>(dir + "/depth.bin", (size_t)W * H);
    const auto member = slurp<int32_t>(dir + "/member.bin", (size_t)(W / 2) * (H / 2));
    const auto pose = slurp<float>(dir + "/pose.bin", 16);
    auto local = slurp<msl_surfel>(dir + "/local.bin", nLocal);

    // ---- ORBextractor::ORBextractor + operator() ----
    msl_orb *orb = msl_orb_create(1000, 1.2f, 8, 20, 7, W, H, 1, 0);
    if (!orb) { std::fprintf(stderr, "msl_orb_create: %s\n", msl_last_error()); return 1; }
    const int cap = msl_orb_capacity(orb), L = msl_orb_levels(orb);
    std::vector<msl_keypoint> kps(cap);
    std::vector<uint8_t> desc((size_t)cap * 32);
    int n = 0;
    CHECK(msl_orb_extract(orb, gray.data(), W, H, (size_t)W, kps.data(), desc.data(), cap, &n));
    std::vector<float> tables(4 * (size_t)L);
    CHECK(msl_orb_scale_tables(orb, &tables[0], &tables[L], &tables[2 * L], &tables[3 * L]));
    msl_orb_destroy(orb);
    dump(dir + "/kps.bin", kps.data(), (size_t)n);
    dump(dir + "/desc.bin", desc.data(), (size_t)n * 32);
    dump(dir + "/tables.bin", tables.data(), tables.size());

    // ---- SurfelFusion::SurfelFusion + fuseInitializeMap (host-vector mode) ----
    msl_sf *sf = msl_sf_create(W, H, fx, fy, cx, cy, 30.0f, 0.5f, 0);
    if (!sf) { std::fprintf(stderr, "msl_sf_create: %s\n", msl_last_error()); return 1; }
    std::vector<msl_surfel> fresh((size_t)(W / 8) * (H / 8));
    size_t nNew = 0;
    CHECK(msl_sf_fuse(sf, ref, gray.data(), (size_t)W, depth.data(), (size_t)W * 4, member.data(), (size_t)(W / 2) * 4, pose.data(),
                      local.data(), local.size(), fresh.data(), fresh.size(), &nNew));
    msl_sf_destroy(sf);
    dump(dir + "/local_out.bin", local.data(), local.size());
    dump(dir + "/new.bin", fresh.data(), nNew);
    std::printf("host_abi ok: %d keypoints, %zu new surfels\n", n, nNew);
    return 0;
}
