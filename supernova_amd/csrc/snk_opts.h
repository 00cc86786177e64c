// snk_opts.h -- per-context options (round 6): every choice the library makes by itself, and every hook its tests use, is a named option of
// the CONTEXT, set through the C ABI (snk_ctx_set_tuning for the documented knobs, snk_ctx_set_option by name; SNK_TUNING="name=value,..."
// is read once when a context is created, for shell tools).  Nothing in the product path reads the process environment any more except
// the tracing switches (SNK_SYNC_TRACE, SNK_ARENA_TRACE, SNK_ARENA_POISON, SNK_INGEST_TRACE, SNK_HBV_DEPTH), SNK_RCCL_LIB and the host
// decoder's two (SNK_FASTH_LIBDEFLATE, SNK_FASTH_WHOLE_MAX_MB: no context there).
// A top-level entry point makes its context's options the calling thread's (snk_enter); the stages look options up by name.
#pragma once
#include <stdint.h>

constexpr int SNK_MAX_OPTS = 96;
struct snk_opts {
    long long v[SNK_MAX_OPTS];
    bool set[SNK_MAX_OPTS];
};
struct snk_opt_def { const char* name; const char* doc; };
extern const snk_opt_def snk_opt_defs[];
int snk_opt_count();
int snk_opt_index(const char* name);                 // -1: no such option
void snk_opts_init(snk_opts* o);                     // nothing set
int snk_opts_parse(snk_opts* o, const char* text, char* bad, unsigned badcap);      // "name=value,name=value"; 0 ok, else the offending item in `bad`
void snk_opts_enter(const snk_opts* o);              // the calling thread's options from here on (NULL: none set)
// look-ups of the calling thread's options
uint32_t snk_opt_u32(const char* name, uint32_t dflt);
unsigned long long snk_opt_u64(const char* name, unsigned long long dflt);
bool snk_opt_is_set(const char* name);
