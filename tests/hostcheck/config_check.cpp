// Host-side check of rafft_amd/csrc/rafft_config.h (round 5): defaults, parsing, equality of snapshots.  Test infrastructure.
#include <cstdio>
#include <cstdlib>
#include "../../rafft_amd/csrc/rafft_config.h"
static int fails = 0;
#define CHECK(c) do { if (!(c)) { fails++; fprintf(stderr, "FAIL line %d: %s\n", __LINE__, #c); } } while (0)
int main()
{
    const char *vars[] = {"RAFFT_TRACE", "RAFFT_SMALL", "RAFFT_MERGE_SEQS", "RAFFT_EST", "RAFFT_NO_HARVEST", "RAFFT_TEST_OVF_AT", "RAFFT_RL_CAP", "RAFFT_C3_SWITCH",
                          "RAFFT_SPANS", "RAFFT_PROD", "RAFFT_STEP_AHEAD", "RAFFT_SPLIT", "RAFFT_LINGER_US", "RAFFT_BIG_WAVE_FRAC"};
    for (const char *v : vars) unsetenv(v);
    const Config d = read_config();
    CHECK(d.trace == 0 && d.spans == -1 && d.prod == 1 && d.small_n4 == 16 && d.small_n5 == 32 && d.merge_seqs == 16384 && d.max_waves == 3);
    CHECK(d.linger_us == 600 && d.direct_n == 1024 && d.c3_direct == 1 && d.c3_switch == -1 && d.split == -1 && d.step_ahead == 0 && d.mat4 == 1);
    CHECK(d.test_ovf_at == -1 && d.test_hard_fail == -1 && d.rl_cap == -1 && d.est == 0.0 && d.no_harvest == 0 && d.reserve_frac == 0.10);
    CHECK(same_config(d, read_config()));
    setenv("RAFFT_TRACE", "", 1);                 // set to anything: at least the summaries
    CHECK(read_config().trace == 1);
    setenv("RAFFT_TRACE", "3", 1); setenv("RAFFT_SMALL", "8,24", 1); setenv("RAFFT_MERGE_SEQS", "4000", 1); setenv("RAFFT_NO_HARVEST", "0", 1);
    setenv("RAFFT_PROD", "0", 1); setenv("RAFFT_SPLIT", "0", 1); setenv("RAFFT_BIG_WAVE_FRAC", "0.25", 1); setenv("RAFFT_SPANS", "2", 1);
    const Config e = read_config();
    CHECK(e.trace == 3 && e.small_n4 == 8 && e.small_n5 == 24 && e.merge_seqs == 4000 && e.no_harvest == 1 && e.prod == 0 && e.split == 0 && e.big_wave_frac == 0.25 && e.spans == 2);
    CHECK(!same_config(d, e));
#ifdef RAFFT_NO_TEST_HOOKS
    setenv("RAFFT_TEST_OVF_AT", "3", 1); setenv("RAFFT_EST", "2.5", 1); setenv("RAFFT_RL_CAP", "0", 1);
    const Config h = read_config();
    CHECK(h.test_ovf_at == -1 && h.est == 0.0 && h.rl_cap == -1);      // compiled out
#else
    setenv("RAFFT_TEST_OVF_AT", "3", 1); setenv("RAFFT_EST", "2.5", 1); setenv("RAFFT_RL_CAP", "0", 1);
    const Config h = read_config();
    CHECK(h.test_ovf_at == 3 && h.est == 2.5 && h.rl_cap == 0);
#endif
    printf("config: %d failures\n", fails);
    return fails ? 1 : 0;
}
