/* pyints.c -- CPython helper of the host shim: Python ints <-> little-endian 64-bit limb arrays.
 *
 * The reference marshals every scalar through pyo3's BigUint extraction (`Fr::from(BigUint)`, src/bn254/curve.rs:358-361;
 * negative ints raise OverflowError there).  The reference-shaped API of this package (prove(list[int], list[int])) has to
 * do the same for 2^20 and more witness values per call; `int.to_bytes` per element costs ~180 ms at 2^20, one thread of this loop ~45, eight ~8.
 * Pure marshalling: no field arithmetic happens here (values >= the modulus take the Python-level `%`).
 */
#define _GNU_SOURCE
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <sched.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

/* value of a non-negative PyLong into `words` little-endian 64-bit limbs, straight from its 30-bit digits (CPython 3.8 .. 3.11
 * layout; anything else goes through _PyLong_AsByteArray).  Returns 0 ok, 1 does not fit, -1 not applicable. */
static int pack_digits(PyObject* v, unsigned char* dst, size_t words) {
#if PY_VERSION_HEX < 0x030C0000 && PYLONG_BITS_IN_DIGIT == 30
    const PyLongObject* lv = (const PyLongObject*)v;
    const Py_ssize_t nd = Py_SIZE(lv);
    uint64_t out[16];
    if (nd < 0 || words > 16) return -1;
    for (size_t w = 0; w < words; ++w) out[w] = 0;
    size_t bit = 0;
    for (Py_ssize_t i = 0; i < nd; ++i, bit += 30) {
        const uint64_t d = lv->ob_digit[i];
        const size_t w = bit >> 6, o = bit & 63;
        if (w >= words) {
            if (d) return 1;
            continue;
        }
        out[w] |= d << o;
        if (o > 34) {
            const uint64_t hi = d >> (64 - o);
            if (w + 1 < words) out[w + 1] |= hi;
            else if (hi) return 1;
        }
    }
    memcpy(dst, out, words * 8);
    return 0;
#else
    (void)v; (void)dst; (void)words;
    return -1;
#endif
}

/* ---- the bulk of a witness: exact, non-negative ints that fit and are below the modulus -------------------------------
 * Worker threads repack those straight from the digits while the calling thread keeps the GIL (so nothing can mutate the
 * sequence or its ints meanwhile; the workers only read, they touch no reference count).  Everything else -- other
 * types, negative or oversized values, values >= the modulus -- is flagged and handled afterwards in index order by the
 * general path below, so the first offending element still decides the exception. */
typedef struct {
    PyObject** items;
    Py_ssize_t begin, end;
    size_t words;
    unsigned char* dst;
    const uint64_t* mod;   /* NULL = no modulus */
    unsigned char* slow;   /* per element: 1 = leave to the general path */
    int any_slow;          /* out: some element of [begin, end) was flagged */
} PackJob;

static void* pack_worker(void* arg) {
    PackJob* job = (PackJob*)arg;
    const size_t nbytes = job->words * 8;
    int any = 0;
    for (Py_ssize_t i = job->begin; i < job->end; ++i) {
        /* the walk is bound by the latency of the int objects (64 MB of them at 2^20, in allocation order at best): ask for
         * both cache lines a 254-bit int can straddle, well ahead */
        if (i + 24 < job->end) {
            const char* nxt = (const char*)job->items[i + 24];
            __builtin_prefetch(nxt);
            __builtin_prefetch(nxt + 56);
        }
        PyObject* item = job->items[i];
        unsigned char* p = job->dst + (size_t)i * nbytes;
        int ok = PyLong_CheckExact(item) && pack_digits(item, p, job->words) == 0;
        if (ok && job->mod) {
            uint64_t v[16];
            memcpy(v, p, nbytes);
            int cmp = 0;
            for (Py_ssize_t k = (Py_ssize_t)job->words - 1; k >= 0 && cmp == 0; --k) cmp = v[k] > job->mod[k] ? 1 : (v[k] < job->mod[k] ? -1 : 0);
            ok = cmp < 0;
        }
        job->slow[i] = (unsigned char)!ok;
        any |= !ok;
    }
    job->any_slow = any;
    return NULL;
}

#define PACK_MAX_THREADS 32
#define PACK_PARALLEL_MIN 16384

/* worker threads of one call: the cores this process may run on (its affinity mask, not the machine's core count: a
 * container or a GPU box share sees 256 cores and owns 16), at most PACK_MAX_THREADS; ZKMI_PACK_THREADS overrides */
static int pack_threads(void) {
    static int cached = 0;
    if (cached) return cached;
    long t = 0;
    const char* e = getenv("ZKMI_PACK_THREADS");
    if (e && *e) t = atol(e);
    if (t <= 0) {
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) t = CPU_COUNT(&set);
        else t = sysconf(_SC_NPROCESSORS_ONLN);
        if (t > 16) t = 16;
    }
    if (t < 1) t = 1;
    if (t > PACK_MAX_THREADS) t = PACK_MAX_THREADS;
    cached = (int)t;
    return cached;
}

/* Persistent worker pool: creating and joining 16 threads costs ~0.4 ms per call, a quarter of the conversion of a 2^20
 * witness.  The workers sleep on a condition variable between calls; a call hands each of them one PackJob (generation
 * counter) and waits for the count of finished jobs.  Calls are serialised by the GIL (the caller holds it throughout). */
static struct {
    pthread_mutex_t mu;
    pthread_cond_t wake, done;
    pthread_t tids[PACK_MAX_THREADS];
    PackJob* jobs;          /* jobs[1 .. n_jobs-1] belong to the workers 1 .. n_jobs-1 */
    int n_workers;          /* threads created (worker ids 1 .. n_workers) */
    int n_jobs, finished;
    unsigned long generation;
    pid_t owner;            /* a forked child starts without the threads: it builds its own pool */
} g_pool = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, {0}, NULL, 0, 0, 0, 0, 0};

static void* pool_main(void* arg) {
    const int id = (int)(intptr_t)arg;
    unsigned long seen = 0;
    pthread_mutex_lock(&g_pool.mu);
    for (;;) {
        while (g_pool.generation == seen) pthread_cond_wait(&g_pool.wake, &g_pool.mu);
        seen = g_pool.generation;
        if (id < g_pool.n_jobs) {
            PackJob* job = &g_pool.jobs[id];
            pthread_mutex_unlock(&g_pool.mu);
            pack_worker(job);
            pthread_mutex_lock(&g_pool.mu);
            if (++g_pool.finished == g_pool.n_jobs - 1) pthread_cond_signal(&g_pool.done);
        }
    }
    return NULL;
}

/* make sure workers 1 .. want-1 exist; returns the number of jobs that can run in parallel (>= 1) */
static int pool_ensure(int want) {
    if (g_pool.owner != getpid()) {   /* first use, or a forked child: the parent's threads do not exist here */
        pthread_mutex_init(&g_pool.mu, NULL);
        pthread_cond_init(&g_pool.wake, NULL);
        pthread_cond_init(&g_pool.done, NULL);
        g_pool.n_workers = 0;
        g_pool.generation = 0;
        g_pool.owner = getpid();
    }
    while (g_pool.n_workers < want - 1) {
        pthread_attr_t attr;
        pthread_attr_init(&attr);
        pthread_attr_setdetachstate(&attr, PTHREAD_CREATE_DETACHED);
        const int id = g_pool.n_workers + 1;
        const int rc = pthread_create(&g_pool.tids[id], &attr, pool_main, (void*)(intptr_t)id);
        pthread_attr_destroy(&attr);
        if (rc != 0) break;
        g_pool.n_workers = id;
    }
    return g_pool.n_workers + 1;
}

/* elements [begin, end) of items; returns 1 when some element was left to the general path */
static int pack_fast(PyObject** items, Py_ssize_t begin, Py_ssize_t end, size_t words, unsigned char* dst, const uint64_t* mod, unsigned char* slow) {
    const Py_ssize_t n = end - begin;
    int threads = n >= PACK_PARALLEL_MIN ? pack_threads() : 1;
    if (threads > 1) threads = pool_ensure(threads) < threads ? g_pool.n_workers + 1 : threads;
    PackJob jobs[PACK_MAX_THREADS];
    const Py_ssize_t chunk = (n + threads - 1) / threads;
    for (int t = 0; t < threads; ++t) {
        Py_ssize_t b = begin + t * chunk, e = b + chunk > end ? end : b + chunk;
        if (b > end) b = end;
        jobs[t] = (PackJob){items, b, e, words, dst, mod, slow, 0};
    }
    if (threads > 1) {
        pthread_mutex_lock(&g_pool.mu);
        g_pool.jobs = jobs;
        g_pool.n_jobs = threads;
        g_pool.finished = 0;
        ++g_pool.generation;
        pthread_cond_broadcast(&g_pool.wake);
        pthread_mutex_unlock(&g_pool.mu);
    }
    pack_worker(&jobs[0]);
    if (threads > 1) {
        pthread_mutex_lock(&g_pool.mu);
        while (g_pool.finished < threads - 1) pthread_cond_wait(&g_pool.done, &g_pool.mu);
        g_pool.n_jobs = 0;
        pthread_mutex_unlock(&g_pool.mu);
    }
    int any = 0;
    for (int t = 0; t < threads; ++t) any |= jobs[t].any_slow;
    return any;
}

/* ints_to_limbs(seq, words, modulus_or_None, out[, begin, end]) -> None.  Elements [begin, end) of seq (default: all) go to
 * rows [begin, end) of out, a writable buffer of len(seq) * words * 8 bytes: a caller can convert a long list chunk by
 * chunk -- and ship every chunk while the next one is converted -- without slicing the list (a slice copies the pointers
 * and touches every reference count). */
static PyObject* ints_to_limbs(PyObject* self, PyObject* args) {
    PyObject *seq, *modulus;
    Py_ssize_t words, begin = 0, end = -1;
    Py_buffer out;
    if (!PyArg_ParseTuple(args, "OnOw*|nn", &seq, &words, &modulus, &out, &begin, &end)) return NULL;
    PyObject* fast = PySequence_Fast(seq, "expected a sequence of ints");
    if (!fast) { PyBuffer_Release(&out); return NULL; }
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
    const size_t nbytes = (size_t)words * 8;
    unsigned char* dst = (unsigned char*)out.buf;
    unsigned char mod_bytes[128];
    uint64_t mod_words[16];
    int have_mod = modulus != Py_None;
    PyObject* result = NULL;
    PyObject* mod_obj = NULL;   /* the modulus as an exact int (new reference) */
    unsigned char* slow = NULL;
    if (end < 0 || end > n) end = n;
    if (begin < 0) begin = 0;
    if (begin > end) begin = end;
    if ((size_t)out.len < (size_t)n * nbytes || nbytes > sizeof(mod_bytes) || words < 1) {
        PyErr_SetString(PyExc_ValueError, "output buffer too small");
        goto done;
    }
    if (have_mod) {
        /* any integer-like modulus (int, numpy integer) through __index__; a non-integer raises TypeError as the pure-Python
         * path does, and a modulus below 1 is rejected instead of being read as unsigned bytes */
        mod_obj = PyNumber_Index(modulus);
        if (!mod_obj) goto done;
        if (_PyLong_Sign(mod_obj) <= 0) {
            PyErr_SetString(PyExc_ValueError, "modulus must be a positive integer");
            goto done;
        }
        if (_PyLong_AsByteArray((PyLongObject*)mod_obj, mod_bytes, nbytes, 1, 0) < 0) goto done;
    }
    slow = (unsigned char*)malloc(n > 0 ? (size_t)n : 1);
    if (!slow) { PyErr_NoMemory(); goto done; }
    memcpy(mod_words, mod_bytes, have_mod ? nbytes : 0);
    if (!pack_fast(PySequence_Fast_ITEMS(fast), begin, end, (size_t)words, dst, have_mod ? mod_words : NULL, slow)) begin = end;  /* nothing flagged */
    for (Py_ssize_t i = begin; i < end; ++i) {
        if (!slow[i]) continue;
        PyObject* item = PySequence_Fast_GET_ITEM(fast, i);
        PyObject* v = PyNumber_Index(item);  /* accepts numpy integers too; new reference */
        if (!v) goto done;
        if (_PyLong_Sign(v) < 0) {
            Py_DECREF(v);
            PyErr_SetString(PyExc_OverflowError, "can't convert negative int to unsigned");
            goto done;
        }
        unsigned char* p = dst + (size_t)i * nbytes;
        int need_reduce = 0;
        int packed = pack_digits(v, p, (size_t)words);
        if (packed == 1 && !have_mod) {
            Py_DECREF(v);
            PyErr_SetString(PyExc_OverflowError, "int too big to convert");
            goto done;
        }
        if (packed == 1) {
            need_reduce = 1;
        } else if (packed < 0 && _PyLong_AsByteArray((PyLongObject*)v, p, nbytes, 1, 0) < 0) {
            if (!have_mod || !PyErr_ExceptionMatches(PyExc_OverflowError)) { Py_DECREF(v); goto done; }
            PyErr_Clear();
            need_reduce = 1;
        } else if (have_mod) {
            /* v >= modulus ?  compare from the most significant byte */
            int cmp = 0;
            for (Py_ssize_t k = (Py_ssize_t)nbytes - 1; k >= 0 && cmp == 0; --k) cmp = (int)p[k] - (int)mod_bytes[k];
            need_reduce = cmp >= 0;
        }
        if (need_reduce) {
            PyObject* r = PyNumber_Remainder(v, mod_obj);
            if (!r) { Py_DECREF(v); goto done; }
            int rc = _PyLong_AsByteArray((PyLongObject*)r, p, nbytes, 1, 0);
            Py_DECREF(r);
            if (rc < 0) { Py_DECREF(v); goto done; }
        }
        Py_DECREF(v);
    }
    Py_INCREF(Py_None);
    result = Py_None;
done:
    free(slow);
    Py_XDECREF(mod_obj);
    Py_DECREF(fast);
    PyBuffer_Release(&out);
    return result;
}

/* limbs_to_ints(buffer, words) -> list[int] */
static PyObject* limbs_to_ints(PyObject* self, PyObject* args) {
    Py_buffer in;
    Py_ssize_t words;
    if (!PyArg_ParseTuple(args, "y*n", &in, &words)) return NULL;
    const size_t nbytes = (size_t)words * 8;
    PyObject* list = NULL;
    if (nbytes == 0 || (size_t)in.len % nbytes) {
        PyErr_SetString(PyExc_ValueError, "buffer length is not a multiple of the element size");
        goto done;
    }
    {
        const Py_ssize_t n = (Py_ssize_t)((size_t)in.len / nbytes);
        list = PyList_New(n);
        if (!list) goto done;
        const unsigned char* src = (const unsigned char*)in.buf;
        for (Py_ssize_t i = 0; i < n; ++i) {
            PyObject* v = _PyLong_FromByteArray(src + (size_t)i * nbytes, nbytes, 1, 0);
            if (!v) { Py_CLEAR(list); goto done; }
            PyList_SET_ITEM(list, i, v);
        }
    }
done:
    PyBuffer_Release(&in);
    return list;
}

static PyMethodDef methods[] = {
    {"ints_to_limbs", ints_to_limbs, METH_VARARGS, "ints_to_limbs(seq, words, modulus_or_None, out_buffer[, begin, end])"},
    {"limbs_to_ints", limbs_to_ints, METH_VARARGS, "limbs_to_ints(buffer, words) -> list of ints"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_pyints", "Python int <-> limb array marshalling", -1, methods};

PyMODINIT_FUNC PyInit__pyints(void) { return PyModule_Create(&module); }
