// pm_device.h -- device code of the red-black PatchMatch path for gfx950 (MI355X).
//
// Written for CDNA4 directly: 64-wide wavefronts, one lane per pixel of the active checkerboard
// colour, the reference-image tile staged in LDS, camera constants wave-uniform (scalar loads),
// source views read with software bilinear filtering (gfx950 has no image instructions,
// SURVEY.md F1).  No MFMA: the patch cost is a small stencil reduction (BASELINE.json).
//
// The arithmetic follows the numerical model M1-M4 of DESIGN.md section 3 (fp32 lerp bilinear
// taps, exp_model, x*(1/z), explicit fmaf); build with -ffp-contract=off so only the fmaf()
// written here fuse.  The reference functions each piece stands for are cited by file:line of
// reference gipuma.cu.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pm_core.h"       // problem block, numerical model, geometry, packed source views
#include "pm_cost.h"       // patch costs and their multi-view combination
#include "pm_prefilter.h"  // lower-bound prefilter of refinement candidates
#include "pm_sweep.h"      // init / sweep / finalize kernels
