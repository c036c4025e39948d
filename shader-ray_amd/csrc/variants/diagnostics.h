// variants/diagnostics.h -- the diagnostic build (-DSHRAY_DIAGNOSTICS, `make diag`; profiles/timeline.py, leaf_stage_histogram.py):
// wave-level tallies threaded through the stages of wave_traversal.h / leaf_stage.h.  Not part of the product build: wave_traversal.h
// defines the same macros empty.
#pragma once

// Diagnostic build only: wave-level tallies {node-loop iterations, leaf-loop iterations,
// cycles in the node loop, cycles in the leaf loop}, read by profiles/timeline.py.

// SHRAY_DIAG_KHIST (with SHRAY_DIAGNOSTICS; profiles/leaf_stage_histogram.py): the eight tallies are instead a histogram of
// the dealt leaf stages by the number of parked lanes K -- bins K = 1, 2, 3-4, 5-8, 9-16, 17-32, > 32 (the plain loop) --,
// each word {stages, bits 0-23; rounds of three strided fetches the stage runs, bits 24-43; 16-byte-per-lane fetches a
// stage would run if every group fetched its leaf's bytes as consecutive chunks, bits 44-63}; nothing is timed.
// SHRAY_DIAG_UNIFORM (with SHRAY_DIAGNOSTICS; profiles/uniform_visit_histogram.py, round 6): the eight tallies count the node stage's
// wave-visits -- {all, wave-uniform (every walking lane at one record), of those: every lane enters, no lane enters, mixed; visits
// with two distinct records, with three or four; walking lanes in all visits} (variants/diag_uniform_visit.inc).
#if defined(SHRAY_DIAGNOSTICS) && (defined(SHRAY_DIAG_KHIST) || defined(SHRAY_DIAG_UNIFORM))
#ifndef SHRAY_DIAG_KHIST_FROM
#define SHRAY_DIAG_KHIST_FROM 32
#endif
#define SHRAY_DIAG_DECL unsigned long long diag_tally[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define SHRAY_DIAG_T0
#define SHRAY_DIAG_WAIT(k) ((void)0)
#define SHRAY_DIAG_COUNT(k) ((void)0)
#define SHRAY_DIAG_PARAM , unsigned long long *diag_tally_ref
#define SHRAY_DIAG_ARG , diag_tally
#define SHRAY_DIAG_ARG_FWD , diag_tally_ref
#elif defined(SHRAY_DIAGNOSTICS)
#define SHRAY_DIAG_DECL unsigned long long diag_tally[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define SHRAY_DIAG_T0 const unsigned long long diag_t0 = __builtin_amdgcn_s_memtime();
#define SHRAY_DIAG_WAIT(k) do { __builtin_amdgcn_s_waitcnt(0); diag_tally_ref[k] += __builtin_amdgcn_s_memtime() - diag_t0; } while (0)
#define SHRAY_DIAG_COUNT(k) (diag_tally_ref[k]++)
#define SHRAY_DIAG_PARAM , unsigned long long *diag_tally_ref
#define SHRAY_DIAG_ARG , diag_tally
#define SHRAY_DIAG_ARG_FWD , diag_tally_ref
#endif
