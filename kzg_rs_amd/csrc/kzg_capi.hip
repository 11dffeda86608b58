// kzg_capi.hip - libkzg_rs_amd.so: the C ABI of include/kzg_rs_amd.h over the gfx950 kernels.
//
// Host side of the reference's verification flow (src/kzg_proof.rs:353-525).  The host does no
// field or curve arithmetic: it parses bytes, launches kernels, hashes the (n * 160 + 32)-byte
// batch transcript (src/kzg_proof.rs:291-348) with SHA-256 - a single serial chain that a host
// core finishes in a fraction of the time one GPU lane would need - and reduces the 256-bit
// digest below r by at most two subtractions.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/kzg_rs_amd.h"
#include "fr_kernels.hpp"
#include "g1.hpp"
#include "msm.hpp"
#include "msm_fixed.hpp"
#include "slp.hpp"
#include "slp2.hpp"
#include "proof_kernels.hpp"

using namespace kzg;

// ---------------------------------------------------------------- embedded SLP programs
#if !defined(__HIP_DEVICE_COMPILE__)
// KZG_DATA_DIR is a string literal supplied by the build (kzg_rs_amd/build.py)
__asm__(".section .rodata\n"
        ".balign 16\n.global kzg_slp_prep_begin\nkzg_slp_prep_begin:\n.incbin \"" KZG_DATA_DIR "/slp_prep.bin\"\n"
        ".global kzg_slp_prep_end\nkzg_slp_prep_end:\n"
        ".balign 16\n.global kzg_slp_verify_begin\nkzg_slp_verify_begin:\n.incbin \"" KZG_DATA_DIR "/slp_verify.bin\"\n"
        ".global kzg_slp_verify_end\nkzg_slp_verify_end:\n"
        ".balign 16\n.global kzg_slp_verify2_begin\nkzg_slp_verify2_begin:\n.incbin \"" KZG_DATA_DIR "/slp_verify2.bin\"\n"
        ".global kzg_slp_verify2_end\nkzg_slp_verify2_end:\n"
        ".balign 16\n.global kzg_slp_scalars_begin\nkzg_slp_scalars_begin:\n.incbin \"" KZG_DATA_DIR "/slp_scalars.bin\"\n"
        ".global kzg_slp_scalars_end\nkzg_slp_scalars_end:\n"
        ".balign 16\n.global kzg_slp_verify3_begin\nkzg_slp_verify3_begin:\n.incbin \"" KZG_DATA_DIR "/slp_verify3.bin\"\n"
        ".global kzg_slp_verify3_end\nkzg_slp_verify3_end:\n"
        ".balign 16\n.global kzg_fixed_base_begin\nkzg_fixed_base_begin:\n.incbin \"" KZG_DATA_DIR "/fixed_base.bin\"\n"
        ".global kzg_fixed_base_end\nkzg_fixed_base_end:\n"
        ".text\n");
#endif
extern "C" const unsigned char kzg_slp_prep_begin[], kzg_slp_prep_end[], kzg_slp_verify_begin[], kzg_slp_verify_end[], kzg_slp_verify2_begin[],
    kzg_slp_verify2_end[], kzg_slp_scalars_begin[], kzg_slp_scalars_end[], kzg_slp_verify3_begin[], kzg_slp_verify3_end[], kzg_fixed_base_begin[],
    kzg_fixed_base_end[];

// The library is ONE translation unit (device code and the host ABI share types and inline helpers); its parts, in
// dependency order:
#include "capi_glue_kernels.hpp"
#include "capi_host_util.hpp"
#include "capi_settings.hpp"
#include "capi_verify.hpp"
#include "capi_multi.hpp"
#include "capi_coalesce.hpp"
#include "capi_pipeline.hpp"
#include "capi_pieces.hpp"
#include "capi_prover.hpp"
#include "capi_debug.hpp"
