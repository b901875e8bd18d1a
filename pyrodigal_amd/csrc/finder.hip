// Finder-level entry point (whole-batch GeneFinder.find_genes drop-in).  Placeholder until the
// extraction / scoring kernels land: fails loudly instead of computing anything on the CPU.
#include "pga_internal.h"

void pga_finder_release(pga_ctx*) {}
int pga_finder_models_changed(pga_ctx*) { return PGA_OK; }

extern "C" int pga_find_genes_batch(pga_ctx* c, int32_t, const char* const*, const int64_t*, const pga_params*, pga_result** out) {
    if (out) *out = nullptr;
    if (c) c->err = "pga_find_genes_batch: not implemented yet";
    return PGA_EDEVICE;
}
extern "C" void pga_result_free(pga_result*) {}
