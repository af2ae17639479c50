/* Debug aid (tests/conftest.py, EGOMI_ABORT_TRACE=1): a SIGABRT handler that writes the NATIVE backtrace of the thread that called abort() to stderr and
 * then hands over to the handler that was installed before (pytest's faulthandler).  Built on the spot:  gcc -shared -fPIC -O1 -o <out>.so abort_trace.c */
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static struct sigaction prev_;

static void on_abort(int sig, siginfo_t* si, void* uc) {
    void* bt[96];
    const int n = backtrace(bt, 96);
    static const char msg[] = "\n=== native backtrace of the thread that raised SIGABRT ===\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(bt, n, 2);
    if ((prev_.sa_flags & SA_SIGINFO) && prev_.sa_sigaction) prev_.sa_sigaction(sig, si, uc);
    else if (!(prev_.sa_flags & SA_SIGINFO) && prev_.sa_handler != SIG_DFL && prev_.sa_handler != SIG_IGN && prev_.sa_handler) prev_.sa_handler(sig);
    signal(sig, SIG_DFL);
    raise(sig);
}

/* may be called again at any time (somebody may have replaced the handler in between): the handler found in place becomes the one chained to */
void install_abort_trace(void) {
    struct sigaction sa, old;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_abort;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER;
    sigaction(SIGABRT, &sa, &old);
    if (!((old.sa_flags & SA_SIGINFO) && old.sa_sigaction == on_abort)) prev_ = old;
}
