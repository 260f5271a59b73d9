// LD_PRELOAD helper: print a native backtrace on SIGABRT (debugging aid, not part of the product).
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <unistd.h>
static void handler(int sig) {
	void* frames[64];
	int n = backtrace(frames, 64);
	dprintf(2, "\n=== SIGABRT native backtrace (%d frames) ===\n", n);
	backtrace_symbols_fd(frames, n, 2);
	signal(sig, SIG_DFL);
	raise(sig);
}
__attribute__((constructor)) static void init(void) { signal(SIGABRT, handler); }
