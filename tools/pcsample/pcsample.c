/* Flat PC-sampling profiler for the host side of the driver (development tool, not part of the product library): a process-wide CPU-time timer
 * (ITIMER_PROF) delivers SIGPROF to whichever thread is burning CPU; the handler stores the interrupted program counter.  pcs_stop() writes the
 * samples and /proc/self/maps so that tools/host_prof.py can attribute them to functions with `nm`.
 *   gcc -O2 -fPIC -shared -o libpcsample.so pcsample.c */
#define _GNU_SOURCE
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/time.h>
#include <unistd.h>
#include <ucontext.h>

#define PCS_CAP (1 << 22)
static uint64_t g_pc[PCS_CAP];
static volatile long g_n = 0;
static struct sigaction g_old;

static void on_prof(int sig, siginfo_t* si, void* uc_) {
    (void)sig; (void)si;
    ucontext_t* uc = (ucontext_t*)uc_;
    long i = __atomic_fetch_add(&g_n, 1, __ATOMIC_RELAXED);
    if (i < PCS_CAP) g_pc[i] = (uint64_t)uc->uc_mcontext.gregs[REG_RIP];
}

/* One timer per thread that exists now, each on that thread's own CPU clock and delivering SIGPROF to that thread: samples are proportional to the
 * CPU time of every thread (a process-wide ITIMER_PROF signal mostly lands on the main thread). */
#include <dirent.h>
#include <stdlib.h>
#include <time.h>
#ifndef sigev_notify_thread_id
#define sigev_notify_thread_id _sigev_un._tid
#endif
#define PCS_MAX_THREADS 512
static timer_t g_timers[PCS_MAX_THREADS];
static int g_ntimers = 0;

int pcs_start(int hz) {
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = on_prof;
    sa.sa_flags = SA_SIGINFO | SA_RESTART;
    sigemptyset(&sa.sa_mask);
    if (sigaction(SIGPROF, &sa, &g_old)) return -1;
    g_n = 0;
    g_ntimers = 0;
    DIR* d = opendir("/proc/self/task");
    if (!d) return -2;
    struct dirent* e;
    while ((e = readdir(d)) && g_ntimers < PCS_MAX_THREADS) {
        const int tid = atoi(e->d_name);
        if (tid <= 0) continue;
        const clockid_t cid = ((~(clockid_t)tid) << 3) | 6;   /* MAKE_THREAD_CPUCLOCK(tid, CPUCLOCK_SCHED) */
        struct sigevent sev;
        memset(&sev, 0, sizeof(sev));
        sev.sigev_notify = SIGEV_THREAD_ID;
        sev.sigev_signo = SIGPROF;
        sev.sigev_notify_thread_id = tid;
        timer_t t;
        if (timer_create(cid, &sev, &t)) continue;
        struct itimerspec its;
        its.it_interval.tv_sec = 0; its.it_interval.tv_nsec = 1000000000L / hz;
        its.it_value = its.it_interval;
        if (timer_settime(t, 0, &its, 0)) { timer_delete(t); continue; }
        g_timers[g_ntimers++] = t;
    }
    closedir(d);
    return g_ntimers;
}

long pcs_stop(const char* path) {
    for (int i = 0; i < g_ntimers; i++) timer_delete(g_timers[i]);
    g_ntimers = 0;
    struct timespec ts = {0, 5000000};
    nanosleep(&ts, 0);
    sigaction(SIGPROF, &g_old, 0);
    long n = g_n < PCS_CAP ? g_n : PCS_CAP;
    FILE* f = fopen(path, "w");
    if (!f) return -1;
    FILE* m = fopen("/proc/self/maps", "r");
    char line[1024];
    if (m) { while (fgets(line, sizeof(line), m)) if (strstr(line, " r-xp ") || strstr(line, " r-x")) fprintf(f, "M %s", line); fclose(m); }
    for (long i = 0; i < n; i++) fprintf(f, "S %llx\n", (unsigned long long)g_pc[i]);
    fclose(f);
    return n;
}
