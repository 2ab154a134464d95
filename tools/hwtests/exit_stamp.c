/* exit_stamp.c -- measurement aid of tools/hwtests/exit_cost.py (never part of the product): a wall-clock stamp on stderr from a C
 * atexit handler (registered when the library is loaded: late in the process = early among the handlers) and from the library's
 * destructor (preloaded = the very end of the process's user-space life). */
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#ifndef STAMP_NAME
#define STAMP_NAME "c"
#endif
static void stamp(const char* what)
{
    struct timespec t;
    clock_gettime(CLOCK_REALTIME, &t);
    fprintf(stderr, "exit-trace %s_%s %ld.%06ld\n", STAMP_NAME, what, (long)t.tv_sec, (long)(t.tv_nsec / 1000));
    fflush(stderr);
}
static void at_exit(void) { stamp("atexit"); }
__attribute__((constructor)) static void init(void) { atexit(at_exit); }
__attribute__((destructor)) static void fini(void) { stamp("destructor"); }
