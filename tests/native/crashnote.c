/* Test infrastructure (not product): names the running test when the interpreter dies on a fatal signal.
 *
 * tests/conftest.py installs these handlers UNDER Python's faulthandler: faulthandler dumps the thread stacks, restores
 * the previous handler (this one) and re-raises, so the lines written here are the LAST thing on stderr -- the tail a
 * truncated log keeps.  Async-signal-safe calls only (write, sigaction, raise).
 */
#include <signal.h>
#include <string.h>
#include <unistd.h>

static char g_note[1024];
static volatile int g_len = 0;
static int g_fd = 2;

void osp_crashnote_set(const char *s) {
    size_t n = strlen(s);
    if (n > sizeof(g_note) - 1) n = sizeof(g_note) - 1;
    g_len = 0;
    memcpy(g_note, s, n);
    g_note[n] = 0;
    g_len = (int)n;
}

static void put(const char *s, size_t n) {
    while (n) {
        ssize_t w = write(g_fd, s, n);
        if (w <= 0) return;
        s += w;
        n -= (size_t)w;
    }
}

static void on_fatal(int sig) {
    static const char head[] = "\n[osp-crashnote] fatal signal ";
    char num[4];
    int k = 0;
    if (sig >= 10) num[k++] = (char)('0' + sig / 10);
    num[k++] = (char)('0' + sig % 10);
    put(head, sizeof(head) - 1);
    put(num, (size_t)k);
    put(" while running: ", 16);
    put(g_note, (size_t)g_len);
    put("\n", 1);
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = SIG_DFL;
    sigaction(sig, &sa, 0);
    raise(sig);
}

int osp_crashnote_install(int fd) {
    static const int sigs[] = {SIGABRT, SIGSEGV, SIGBUS, SIGFPE, SIGILL};
    g_fd = fd;
    for (unsigned i = 0; i < sizeof(sigs) / sizeof(sigs[0]); ++i) {
        struct sigaction sa;
        memset(&sa, 0, sizeof(sa));
        sa.sa_handler = on_fatal;
        sa.sa_flags = SA_NODEFER;
        if (sigaction(sigs[i], &sa, 0) != 0) return -1;
    }
    return 0;
}
