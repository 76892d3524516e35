/*
 * ref_prelude.h -- TEST INFRASTRUCTURE ONLY.  Compiled IN FRONT OF the reference translation unit
 * (streamed from /root/reference by build_ref.sh, never copied into this repo).
 *
 * The reference sizes its static buffers with compile-time macros (SemiGlobalMatching.h:14-19).
 * To build it for another capacity without editing its text we include its header once with the
 * six `extern` buffer declarations renamed (so they do not clash with the re-sized definitions in
 * the .c), then re-define the three capacity macros.  MAX_IMG_SIZE / MAX_DISP_IMG_SIZE are defined
 * in terms of those and expand lazily, so they follow.
 */
#define census_right_buffer hdr_only_census_right_buffer
#define census_left_buffer  hdr_only_census_left_buffer
#define cost_init_buffer    hdr_only_cost_init_buffer
#define cost_aggr_buffer    hdr_only_cost_aggr_buffer
#define disp_left_buffer    hdr_only_disp_left_buffer
#define disp_right_buffer   hdr_only_disp_right_buffer
#include "SemiGlobalMatching.h"
#undef census_right_buffer
#undef census_left_buffer
#undef cost_init_buffer
#undef cost_aggr_buffer
#undef disp_left_buffer
#undef disp_right_buffer

#undef MAX_IMG_WIDTH
#undef MAX_IMG_HEIGHT
#undef MAX_DISPARITY_RANGE
#define MAX_IMG_WIDTH       REF_W
#define MAX_IMG_HEIGHT      REF_H
#define MAX_DISPARITY_RANGE REF_D

#include <stddef.h>
/* counts the out-of-image path steps the guard (inserted by build_ref.sh) drops; SURVEY.md Q6 */
static unsigned long ref_oob_dropped;
