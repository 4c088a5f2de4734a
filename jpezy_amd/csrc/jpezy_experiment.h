// jpezy_experiment.h -- one gate for every build switch that yields WRONG RESULTS (timing probes of tools/ab/ab_build.py).
// A library built with any of them must also be built with -DJPEZY_EXPERIMENT: it then says so through
// jpezy_hip_is_experimental_build(), and the Python binding refuses to load it unless JPEZY_ALLOW_EXPERIMENT=1 is set.
// Switches that keep the results right (JPEZY_DEC_LDS_R01, JPEZY_DEC_INTERLEAVE, JPEZY_ENT_NOFAST, JPEZY_TRACE,
// JPEZY_DUMP_T, register / occupancy knobs) are not gated.
#pragma once
#if (defined(JPEZY_ABL_PS_NOLOAD) || defined(JPEZY_ABL_NOSTORE) || defined(JPEZY_ABL_NOGUARD) || defined(JPEZY_ABL_NOCFLAG) || defined(JPEZY_ABL_LIGHT_TAIL) || defined(JPEZY_ENT_ABL)) && !defined(JPEZY_EXPERIMENT)
#error "JPEZY_ABL_* / JPEZY_ENT_ABL produce wrong results: such a build must also define JPEZY_EXPERIMENT"
#endif
