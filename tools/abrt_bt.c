/* LD_PRELOAD helper: native backtrace on SIGABRT / SIGSEGV (the image has no gdb). gcc -shared -fPIC -o tools/abrt_bt.so tools/abrt_bt.c */
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static void handler(int sig) {
    void* frames[64];
    int n = backtrace(frames, 64);
    const char msg[] = "\n==== native backtrace ====\n";
    write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
__attribute__((constructor)) static void install(void) {
    signal(SIGABRT, handler);
    signal(SIGSEGV, handler);
}
