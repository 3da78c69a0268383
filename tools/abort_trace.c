// Debug aid: LD_PRELOAD this to get the native stack of the thread that calls abort() (SIGABRT) — the Python faulthandler only
// shows the Python frames of the main thread.  gcc -shared -fPIC -o build/libaborttrace.so tools/abort_trace.c
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
#include <sys/syscall.h>
static int out_fd = 2;     // the stderr of process start (pytest redirects fd 2 while a test runs)
static void on_abort(int sig) {
    void* frames[96];
    char msg[128];
    int n = backtrace(frames, 96);
    int len = snprintf(msg, sizeof msg, "\n==== SIGABRT in thread %ld: native stack ====\n", (long)syscall(SYS_gettid));
    if (write(out_fd, msg, (size_t)len) < 0) {}
    backtrace_symbols_fd(frames, n, out_fd);
    signal(sig, SIG_DFL);
    raise(sig);
}
__attribute__((constructor)) static void install(void) {
    struct sigaction sa;
    void* warm[4];
    out_fd = dup(2);
    backtrace(warm, 4);          // loads libgcc now, not inside the handler
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_abort;
    sigaction(SIGABRT, &sa, NULL);
}
