// Entry points not implemented yet in this round fail loudly (never a silent CPU fallback).
#include "../../include/tmat.h"
#include "tmat_internal.h"
#define NOTYET(name) do { tmat::set_error(name ": not implemented yet"); return TMAT_E_ARG; } while (0)
extern "C" {
int tmat_segment_batch(tmat_handle, const uint16_t *, int, int, int, float, double *) { NOTYET("tmat_segment_batch"); }
int tmat_postprocess_batch(tmat_handle, const double *, int, int, int, int, float *) { NOTYET("tmat_postprocess_batch"); }
int tmat_analyze_batch_dev(tmat_handle, const uint16_t *, int, int, int, float, int, float, float, int, int, int, int, int64_t,
                           tmat_row *) { NOTYET("tmat_analyze_batch_dev"); }
int tmat_analyze_batch(tmat_handle, const uint16_t *, int, int, int, float, int, float, float, int, int, int, int, int64_t,
                       tmat_row *) { NOTYET("tmat_analyze_batch"); }
}
