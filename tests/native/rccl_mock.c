/* Test double for librccl: the nine entry points libshm_grid.so uses, implemented over POSIX shared memory so that several
 * PROCESSES sharing ONE GPU can exercise the multi-rank code path of the solver (RCCL itself refuses two ranks on one device).
 * Semantics: every operation first drains the caller's stream, then moves data through host shared memory; point-to-point
 * messages are matched per (source, destination) pair in issue order; all-reduce sums in rank order (deterministic).
 * TEST INFRASTRUCTURE ONLY -- selected with the environment variable SHM_RCCL_LIB, never shipped in the product path.
 * Build: gcc -O2 -shared -fPIC tests/native/rccl_mock.c -o <out>.so -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -L/opt/rocm/lib -lamdhip64 -lrt -lpthread */
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#define MAXR 8
#define SLOT_BYTES (8u << 20) /* per (src,dst) mailbox */
#define RED_BYTES (1u << 20)  /* per-rank all-reduce slot */

typedef struct {
    pthread_barrier_t bar;
    volatile uint64_t produced[MAXR][MAXR], consumed[MAXR][MAXR];
    volatile int ready;
} header_t;

typedef struct ncclComm {
    int rank, nranks;
    header_t* h;
    char* mail; /* [src][dst][SLOT_BYTES] */
    char* red;  /* [rank][RED_BYTES] */
    size_t total;
    char name[64];
} comm_t;

typedef struct { char internal[128]; } mock_uid_t;

enum { OP_SEND = 1, OP_RECV = 2 };
typedef struct { int kind, peer; void* buf; size_t bytes; comm_t* c; hipStream_t s; } op_t;
static __thread op_t g_ops[256];
static __thread int g_nops = 0, g_group = 0;

static size_t dtype_size(int dt) { return dt == 8 ? 8 : (dt == 7 ? 4 : 1); }

int ncclGetUniqueId(mock_uid_t* id) {
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/shmmock_%d_%ld", (int)getpid(), (long)time(NULL));
    return 0;
}

int ncclCommInitRank(comm_t** out, int nranks, mock_uid_t id, int rank) {
    if (nranks > MAXR) return 1;
    comm_t* c = (comm_t*)calloc(1, sizeof *c);
    c->rank = rank;
    c->nranks = nranks;
    strncpy(c->name, id.internal, sizeof c->name - 1);
    c->total = sizeof(header_t) + (size_t)MAXR * MAXR * SLOT_BYTES + (size_t)MAXR * RED_BYTES;
    int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return 2;
    if (ftruncate(fd, (off_t)c->total) != 0) return 3;
    void* p = mmap(NULL, c->total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return 4;
    c->h = (header_t*)p;
    c->mail = (char*)p + sizeof(header_t);
    c->red = c->mail + (size_t)MAXR * MAXR * SLOT_BYTES;
    if (rank == 0) {
        pthread_barrierattr_t a;
        pthread_barrierattr_init(&a);
        pthread_barrierattr_setpshared(&a, PTHREAD_PROCESS_SHARED);
        pthread_barrier_init(&c->h->bar, &a, (unsigned)nranks);
        __sync_synchronize();
        c->h->ready = 1;
    } else {
        while (!c->h->ready) usleep(1000);
    }
    pthread_barrier_wait(&c->h->bar);
    *out = c;
    return 0;
}

int ncclCommDestroy(comm_t* c) {
    if (!c) return 0;
    pthread_barrier_wait(&c->h->bar);
    if (c->rank == 0) shm_unlink(c->name);
    munmap((void*)c->h, c->total);
    free(c);
    return 0;
}

static char* slot(comm_t* c, int src, int dst) { return c->mail + ((size_t)src * MAXR + dst) * SLOT_BYTES; }

static int do_send(op_t* o) {
    comm_t* c = o->c;
    if (o->bytes > SLOT_BYTES) { fprintf(stderr, "rccl_mock: message of %zu bytes exceeds the mailbox\n", o->bytes); return 5; }
    while (c->h->produced[c->rank][o->peer] != c->h->consumed[c->rank][o->peer]) usleep(50); /* mailbox busy */
    if (hipMemcpy(slot(c, c->rank, o->peer), o->buf, o->bytes, hipMemcpyDeviceToHost) != hipSuccess) return 6;
    __sync_synchronize();
    c->h->produced[c->rank][o->peer]++;
    return 0;
}
static int do_recv(op_t* o) {
    comm_t* c = o->c;
    while (c->h->produced[o->peer][c->rank] == c->h->consumed[o->peer][c->rank]) usleep(50); /* nothing yet */
    __sync_synchronize();
    if (hipMemcpy(o->buf, slot(c, o->peer, c->rank), o->bytes, hipMemcpyHostToDevice) != hipSuccess) return 7;
    __sync_synchronize();
    c->h->consumed[o->peer][c->rank]++;
    return 0;
}

static int flush_ops(void) {
    int rc = 0;
    if (g_nops == 0) return 0;
    if (hipStreamSynchronize(g_ops[0].s) != hipSuccess) return 8;
    /* sends and receives alternate per peer in issue order; with one message per pair per group (local_slabs == 1) doing all
       sends first cannot deadlock because every mailbox holds one message */
    for (int a = 0; a < g_nops && !rc; a++) if (g_ops[a].kind == OP_SEND) rc = do_send(&g_ops[a]);
    for (int a = 0; a < g_nops && !rc; a++) if (g_ops[a].kind == OP_RECV) rc = do_recv(&g_ops[a]);
    g_nops = 0;
    return rc;
}

int ncclGroupStart(void) { g_group++; return 0; }
int ncclGroupEnd(void) { if (--g_group == 0) return flush_ops(); return 0; }

static int queue_op(int kind, void* buf, size_t count, int dt, int peer, comm_t* c, hipStream_t s) {
    if (g_nops >= 256) return 9;
    g_ops[g_nops++] = (op_t){kind, peer, buf, count * dtype_size(dt), c, s};
    return g_group ? 0 : flush_ops();
}
int ncclSend(const void* buf, size_t count, int dt, int peer, comm_t* c, hipStream_t s) { return queue_op(OP_SEND, (void*)buf, count, dt, peer, c, s); }
int ncclRecv(void* buf, size_t count, int dt, int peer, comm_t* c, hipStream_t s) { return queue_op(OP_RECV, buf, count, dt, peer, c, s); }

int ncclAllReduce(const void* send, void* recv, size_t count, int dt, int op, comm_t* c, hipStream_t s) {
    (void)op;
    const size_t bytes = count * dtype_size(dt);
    if (dt != 8 || bytes > RED_BYTES) { fprintf(stderr, "rccl_mock: all-reduce supports float64 up to %u bytes\n", RED_BYTES); return 10; }
    if (hipStreamSynchronize(s) != hipSuccess) return 8;
    if (hipMemcpy(c->red + (size_t)c->rank * RED_BYTES, send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 6;
    pthread_barrier_wait(&c->h->bar);
    double* acc = (double*)malloc(bytes);
    memset(acc, 0, bytes);
    for (int r = 0; r < c->nranks; r++) {
        const double* v = (const double*)(c->red + (size_t)r * RED_BYTES);
        for (size_t a = 0; a < count; a++) acc[a] += v[a];
    }
    int rc = hipMemcpy(recv, acc, bytes, hipMemcpyHostToDevice) == hipSuccess ? 0 : 7;
    free(acc);
    pthread_barrier_wait(&c->h->bar);
    return rc;
}

const char* ncclGetErrorString(int rc) {
    static __thread char buf[64];
    snprintf(buf, sizeof buf, "rccl_mock error %d", rc);
    return buf;
}
