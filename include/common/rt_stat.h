/* common/rt_stat.h -- per-NN-op timers emitted by the compiler (reference rt_stat.h, rt_stat.c:14-28);
 * prints "[RT_STAT] <name> takes X seconds." (the line scripts/perf.py:233-241 parses). */
#ifndef ACEHIP_COMMON_RT_STAT_H
#define ACEHIP_COMMON_RT_STAT_H
#ifdef __cplusplus
extern "C" {
#endif
void Tm_start(const char* msg);
void Tm_taken(const char* msg);
#ifdef __cplusplus
}
#endif
#endif
