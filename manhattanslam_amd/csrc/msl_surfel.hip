// temporary stub, replaced by the surfel fusion kernels
#include "msl_common.h"
struct msl_sf { int dummy; };
extern "C" {
msl_sf *msl_sf_create(int, int, float, float, float, float, float, float, int) { msl::set_error("not implemented"); return nullptr; }
void msl_sf_destroy(msl_sf *) {}
int msl_sf_fuse(msl_sf *, int, const uint8_t *, size_t, const float *, size_t, const int32_t *, size_t, const float *, msl_surfel *, size_t, msl_surfel *, size_t, size_t *) { return MSL_ERR_INVALID; }
int msl_sf_map_reserve(msl_sf *, size_t) { return MSL_ERR_INVALID; }
int msl_sf_map_upload(msl_sf *, const msl_surfel *, size_t) { return MSL_ERR_INVALID; }
int msl_sf_map_download(msl_sf *, msl_surfel *, size_t, size_t *) { return MSL_ERR_INVALID; }
int msl_sf_map_size(msl_sf *, size_t *) { return MSL_ERR_INVALID; }
int msl_sf_fuse_resident(msl_sf *, int, const uint8_t *, size_t, const float *, size_t, const int32_t *, size_t, msl_mem, const float *) { return MSL_ERR_INVALID; }
int msl_sf_last_counters(msl_sf *, int64_t *) { return MSL_ERR_INVALID; }
int msl_sf_sync(msl_sf *) { return MSL_ERR_INVALID; }
int msl_sf_set_stream(msl_sf *, void *) { return MSL_ERR_INVALID; }
int msl_sf_debug_seeds(msl_sf *, msl_seed *) { return MSL_ERR_INVALID; }
int msl_sf_debug_index(msl_sf *, int32_t *) { return MSL_ERR_INVALID; }
int msl_sf_profile_enable(msl_sf *, int) { return MSL_ERR_INVALID; }
int msl_sf_profile_read(msl_sf *, float *, int32_t *) { return MSL_ERR_INVALID; }
const char *msl_sf_kernel_name(int) { return ""; }
}
