// Diagnostic stamps of the motion kernels.  EMPTY in the shipped build: LFG_STAMP(...) and LFG_STAMP_PHASE(...) swallow their
// arguments, and this header is the only place of the motion files that asks the preprocessor a question.
//     tools/build_variant.sh stamps -DLFG_MOTION_STAMPS                      per-wave timing and counts of every work unit
//     tools/build_variant.sh phases -DLFG_MOTION_STAMPS -DLFG_STAMP_PHASES   ... and the time in batch tests / walks / full evaluations
// The third lfg_motion of such a build prints what the units of that launch did (motion_stamps_report); the stamps themselves
// add some 40 us to a wave that searches in full.
#pragma once

#ifdef LFG_MOTION_STAMPS
#define LFG_STAMP(...) __VA_ARGS__
#else
#define LFG_STAMP(...)
#endif
#if defined(LFG_MOTION_STAMPS) && defined(LFG_STAMP_PHASES)
#define LFG_STAMP_PHASE(...) __VA_ARGS__
#else
#define LFG_STAMP_PHASE(...)
#endif
