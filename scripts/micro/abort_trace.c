/* LD_PRELOAD helper: print the native backtrace of an abort() (debugging aid; gcc -shared -fPIC -o /tmp/abort_trace.so abort_trace.c). */
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static void on_abort(int s) {
    void *b[64];
    int n = backtrace(b, 64);
    backtrace_symbols_fd(b, n, 2);
    _exit(134);
}
__attribute__((constructor)) static void init(void) { signal(SIGABRT, on_abort); signal(SIGSEGV, on_abort); }
