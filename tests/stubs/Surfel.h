// TEST-ONLY declaration (see tests/stubs/README.md) of the host project's `struct Surfel` (56 bytes, 14 four-byte fields):
// lets the adapters' static_assert(sizeof(msl_surfel) == sizeof(Surfel)) be checked here.
#pragma once
struct Surfel {
    float px, py, pz, nx, ny, nz, size, color;
    int r, g, b;
    float weight;
    int updateTimes, lastUpdate;
};
