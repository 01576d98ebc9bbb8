// temporary stubs for entry points not implemented yet
#include "common.hpp"
using namespace asl;
#define NI(name) return fail(ASL_ERR_STATE, #name ": not implemented yet")
extern "C" {
asl_index_t *asl_index_create(int32_t, int32_t, int32_t, int32_t, int32_t) { fail(ASL_ERR_STATE, "ni"); return nullptr; }
void asl_index_free(asl_index_t *) {}
int asl_index_train(asl_index_t *, int64_t, const float *, uint64_t) { NI(train); }
int asl_index_add(asl_index_t *, int64_t, const float *) { NI(add); }
int asl_index_search(asl_index_t *, int32_t, const float *, int32_t, int32_t, float *, int64_t *) { NI(search); }
int asl_index_reset(asl_index_t *) { NI(reset); }
int64_t asl_index_ntotal(const asl_index_t *) { return 0; }
int asl_index_is_trained(const asl_index_t *) { return 0; }
int asl_index_save(const asl_index_t *, const char *) { NI(save); }
asl_index_t *asl_index_load(const char *) { return nullptr; }
int asl_index_set_niter(asl_index_t *, int32_t) { NI(niter); }
int asl_index_info(const asl_index_t *, asl_index_info_t *) { NI(info); }
int asl_index_get_centroids(const asl_index_t *, float *) { NI(x); }
int asl_index_get_codebooks(const asl_index_t *, float *) { NI(x); }
int asl_index_set_trained(asl_index_t *, const float *, const float *) { NI(x); }
int asl_index_get_lists(const asl_index_t *, int32_t *, int32_t *, uint8_t *, float *) { NI(x); }
int asl_index_shard(asl_index_t *, int32_t, int32_t) { NI(x); }
int asl_index_shard_map(const asl_index_t *, int32_t, int32_t *) { NI(x); }
int asl_topk_merge(int32_t, int32_t, int32_t, const float *, const int64_t *, float *, int64_t *) { NI(x); }
int asl_index_coarse(asl_index_t *, int32_t, const float *, int32_t, float *, int32_t *) { NI(x); }
int asl_index_pq_lut(asl_index_t *, int32_t, const float *, float *) { NI(x); }
asl_library_t *asl_library_create(const asl_peaks_t *, const float *, const uint8_t *) { return nullptr; }
void asl_library_free(asl_library_t *) {}
int64_t asl_library_size(const asl_library_t *) { return 0; }
int asl_search_batch(asl_library_t *, asl_index_t *, const asl_peaks_t *, const asl_search_params_t *, int32_t *, double *, int32_t *, int32_t *, uint32_t *, int32_t, int64_t *) { NI(x); }
int asl_window_candidates(asl_library_t *, int32_t, const double *, int32_t, double, int32_t, int32_t *, int64_t *) { NI(x); }
}
