/* debugging aid: LD_PRELOAD this to get a native backtrace on SIGABRT / SIGSEGV (gcc -shared -fPIC -o libabrt.so abrt_trace.c) */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
static void handler(int sig) {
  void* bt[64];
  int n = backtrace(bt, 64);
  const char* m = sig == SIGABRT ? "\n=== SIGABRT native backtrace ===\n" : "\n=== SIGSEGV native backtrace ===\n";
  if (write(2, m, strlen(m)) < 0) {}
  backtrace_symbols_fd(bt, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}
__attribute__((constructor)) static void init(void) { signal(SIGABRT, handler); signal(SIGSEGV, handler); }
