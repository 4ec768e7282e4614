// comm_protocol.hpp -- the gatherv of hit records to one rank, as a protocol over an abstract transport.  PURE HOST C++17.
//
// RCCL has no gatherv: one all-gather of the per-rank counts, then grouped send / recv.  What makes that safe is that a
// rank NEVER leaves the protocol on its own: a rank whose local scan failed, a root that cannot hold the records, a count
// that overflows the offsets -- every such condition is made known to ALL ranks in an exchange every rank takes part in,
// and all of them return an error before any send or receive is posted.  (Otherwise the peers of a failed rank block
// forever in ncclAllGather / ncclSend: a hang instead of an error code.)
//
//   1. exchange_words(count, or kFailed if this rank has a local error)   -- every rank, always
//      any kFailed  -> the failed rank returns its own error, the others SPM_E_PEER
//      offsets overflow (same counts everywhere, so every rank sees it) -> SPM_E_OVERFLOW on all
//   2. the root reserves its receive buffer; exchange_words(0, or 1 if that failed) -- every rank, always
//      root failed  -> its error on the root, SPM_E_PEER elsewhere
//   3. root: own records copied, recv from every rank with records; others: send if they have any; finish.
//
// The product binds the transport to RCCL + HIP (comm.hpp); the self-check binds it to an in-process loopback of `world`
// threads (spm_hip_comm_selftest: worlds of 1..8, every failure above injected, no GPU needed).
#pragma once

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/spm_hip.h"

namespace spm_hip
{

constexpr uint64_t kGathervFailed = ~0ull; // a rank's word in exchange 1 when it cannot contribute

struct gatherv_transport // blocking where it says so; every call returns an spm_status
{
    virtual ~gatherv_transport() = default;
    // all-gather of one word per rank, result on the host when the call returns
    virtual int exchange_words(uint64_t mine, uint64_t *all) = 0;
    virtual int reserve(uint64_t bytes, void **buffer) = 0;                 // root: room for the gathered records
    virtual int copy_own(void *dst, const void *src, uint64_t bytes) = 0;   // root: its own records into place
    virtual int group_begin() = 0;
    virtual int send(const void *src, uint64_t bytes, int peer) = 0;
    virtual int recv(void *dst, uint64_t bytes, int peer) = 0;
    virtual int group_end() = 0;
    virtual int finish() = 0;                                               // returns when the records are in place
};

// host arithmetic of the gatherv: byte offset of every rank's records in the root's buffer, offsets[world] = total
inline int gatherv_plan(const uint64_t *counts, uint32_t world, uint32_t record_bytes, uint64_t *offsets)
{
    if (!counts || !offsets || world == 0 || record_bytes == 0)
        return SPM_E_INVALID;
    uint64_t at = 0;
    for (uint32_t r = 0; r < world; ++r) {
        offsets[r] = at;
        if (counts[r] > (~0ull - at) / record_bytes)
            return SPM_E_OVERFLOW;
        at += counts[r] * record_bytes;
    }
    offsets[world] = at;
    return SPM_OK;
}

// local_error != SPM_OK: this rank has nothing to contribute (its scan failed, its hit buffer overflowed, ..) -- it still
// takes part in the exchanges so that every rank learns of it.
inline int gatherv_protocol(gatherv_transport &T, int rank, int world, int root, int local_error, const void *local,
                            uint64_t n_local, uint32_t record_bytes, const void **records, uint64_t *n_total, uint64_t *counts)
{
    *records = nullptr;
    *n_total = 0;
    std::vector<uint64_t> cnt((size_t)world), off((size_t)world + 1), st((size_t)world);
    int rc = T.exchange_words(local_error != SPM_OK ? kGathervFailed : n_local, cnt.data());
    if (rc != SPM_OK)
        return rc; // (the transport itself failed: there is nobody left to tell)
    for (int r = 0; r < world; ++r)
        if (cnt[r] == kGathervFailed)
            return local_error != SPM_OK ? local_error : SPM_E_PEER;
    rc = gatherv_plan(cnt.data(), (uint32_t)world, record_bytes, off.data());
    if (rc != SPM_OK)
        return rc; // (every rank holds the same counts: all of them return here)
    void *buf = nullptr;
    int root_rc = SPM_OK;
    if (rank == root)
        root_rc = T.reserve(off[world], &buf);
    rc = T.exchange_words(root_rc != SPM_OK ? 1u : 0u, st.data());
    if (rc != SPM_OK)
        return rc;
    for (int r = 0; r < world; ++r)
        if (st[r] != 0)
            return rank == root && root_rc != SPM_OK ? root_rc : SPM_E_PEER;
    if (rank != root) {
        if (n_local) {
            if ((rc = T.group_begin()) != SPM_OK || (rc = T.send(local, n_local * record_bytes, root)) != SPM_OK ||
                (rc = T.group_end()) != SPM_OK)
                return rc;
        }
        return T.finish();
    }
    if (cnt[root] && (rc = T.copy_own((uint8_t *)buf + off[root], local, cnt[root] * record_bytes)) != SPM_OK)
        return rc;
    bool any = false;
    for (int r = 0; r < world; ++r)
        any = any || (r != root && cnt[r]);
    if (any) {
        if ((rc = T.group_begin()) != SPM_OK)
            return rc;
        for (int r = 0; r < world; ++r)
            if (r != root && cnt[r] && (rc = T.recv((uint8_t *)buf + off[r], cnt[r] * record_bytes, r)) != SPM_OK)
                return rc;
        if ((rc = T.group_end()) != SPM_OK)
            return rc;
    }
    if ((rc = T.finish()) != SPM_OK)
        return rc;
    *records = buf;
    *n_total = off[world] / record_bytes;
    if (counts)
        for (int r = 0; r < world; ++r)
            counts[r] = cnt[r];
    return SPM_OK;
}

// ---- in-process loopback: `world` threads, host memory, rendezvous sends -------------------------------------------------
struct loopback_fabric
{
    int world = 1;
    std::mutex m;
    std::condition_variable cv;
    // all-gather rendezvous
    std::vector<uint64_t> words, snapshot;
    int arrived = 0;
    uint64_t generation = 0;
    // point to point: a posted send waits until the matching recv has taken its bytes
    struct message
    {
        const void *src;
        uint64_t bytes;
        int from, to;
        bool taken;
    };
    std::vector<message *> posted;
    std::atomic<bool> timed_out{false};
    std::chrono::milliseconds patience{5000};
};

struct loopback_transport : gatherv_transport
{
    loopback_fabric *F;
    int rank;
    bool fail_reserve = false; // injected: the root cannot allocate
    std::vector<uint8_t> buffer;
    loopback_transport(loopback_fabric *f, int r) : F(f), rank(r) {}
    int exchange_words(uint64_t mine, uint64_t *all) override
    {
        std::unique_lock<std::mutex> g(F->m);
        const uint64_t gen = F->generation;
        F->words[(size_t)rank] = mine;
        if (++F->arrived == F->world) {
            F->arrived = 0;
            ++F->generation;
            std::copy(F->words.begin(), F->words.end(), all);
            // (the others copy before anybody can start the next exchange: a rank enters it only after this one returned
            // on it, and the last of them is the one that bumps the generation)
            F->snapshot = F->words;
            F->cv.notify_all();
            return SPM_OK;
        }
        if (!F->cv.wait_for(g, F->patience, [&]() { return F->generation != gen; })) {
            F->timed_out = true; // a peer never came: the hang the protocol must not produce
            return SPM_E_HIP;
        }
        std::copy(F->snapshot.begin(), F->snapshot.end(), all);
        return SPM_OK;
    }
    int reserve(uint64_t bytes, void **out) override
    {
        if (fail_reserve)
            return SPM_E_NOMEM;
        buffer.assign((size_t)bytes + 1, 0xEE);
        *out = buffer.data();
        return SPM_OK;
    }
    int copy_own(void *dst, const void *src, uint64_t bytes) override
    {
        memcpy(dst, src, (size_t)bytes);
        return SPM_OK;
    }
    int group_begin() override { return SPM_OK; }
    int group_end() override { return SPM_OK; }
    int send(const void *src, uint64_t bytes, int peer) override
    {
        loopback_fabric::message msg{src, bytes, rank, peer, false};
        std::unique_lock<std::mutex> g(F->m);
        F->posted.push_back(&msg);
        F->cv.notify_all();
        if (!F->cv.wait_for(g, F->patience, [&]() { return msg.taken; })) {
            F->posted.erase(std::find(F->posted.begin(), F->posted.end(), &msg));
            F->timed_out = true;
            return SPM_E_HIP;
        }
        return SPM_OK;
    }
    int recv(void *dst, uint64_t bytes, int peer) override
    {
        std::unique_lock<std::mutex> g(F->m);
        loopback_fabric::message *hit = nullptr;
        const bool ok = F->cv.wait_for(g, F->patience, [&]() {
            for (loopback_fabric::message *q : F->posted)
                if (q->from == peer && q->to == rank && !q->taken) {
                    hit = q;
                    return true;
                }
            return false;
        });
        if (!ok) {
            F->timed_out = true;
            return SPM_E_HIP;
        }
        if (hit->bytes != bytes)
            return SPM_E_INVALID; // (sizes must agree, as with ncclSend / ncclRecv)
        memcpy(dst, hit->src, (size_t)bytes);
        hit->taken = true;
        F->posted.erase(std::find(F->posted.begin(), F->posted.end(), hit));
        F->cv.notify_all();
        return SPM_OK;
    }
    int finish() override { return SPM_OK; }
};

// Self-check of the protocol over the loopback.  Rank r contributes (seed + 17 r) % 1000 records of `record_bytes` bytes
// whose content spells (rank, index).  scenario: 0 clean; 1 rank `victim` has a local error (SPM_E_OVERFLOW); 2 the root
// cannot reserve its buffer; 3 the counts overflow the offsets; 4 every rank has no records.
// Returns SPM_OK iff every rank returned what the protocol promises, nothing hung, and (clean runs) the root holds every
// rank's records at the planned offsets.  detail[0..world) = the status every rank returned.
inline int comm_selftest(int world, int root, int scenario, int victim, uint32_t record_bytes, uint64_t seed, int *detail)
{
    if (world < 1 || world > 64 || root < 0 || root >= world || record_bytes == 0 || record_bytes > 64)
        return SPM_E_INVALID;
    if (scenario == 3 && world == 1)
        return SPM_OK; // (one rank's count alone cannot overflow the offsets)
    loopback_fabric F;
    F.world = world;
    F.words.assign((size_t)world, 0);
    F.snapshot.assign((size_t)world, 0);
    F.patience = std::chrono::milliseconds(3000);
    std::vector<int> status((size_t)world, -99);
    std::vector<std::vector<uint8_t>> local((size_t)world);
    std::vector<uint64_t> n((size_t)world);
    for (int r = 0; r < world; ++r) {
        n[(size_t)r] = scenario == 4 ? 0 : (seed + 17ull * (uint64_t)r) % 1000;
        local[(size_t)r].resize((size_t)(n[(size_t)r] * record_bytes + 1));
        for (uint64_t i = 0; i < n[(size_t)r]; ++i)
            for (uint32_t b = 0; b < record_bytes; ++b)
                local[(size_t)r][(size_t)(i * record_bytes + b)] = (uint8_t)(r * 31 + i * 7 + b);
    }
    std::vector<loopback_transport *> T((size_t)world);
    for (int r = 0; r < world; ++r)
        T[(size_t)r] = new loopback_transport(&F, r);
    if (scenario == 2)
        T[(size_t)root]->fail_reserve = true;
    const void *root_records = nullptr;
    uint64_t root_total = 0;
    std::vector<uint64_t> root_counts((size_t)world, 0);
    std::vector<std::thread> th;
    for (int r = 0; r < world; ++r)
        th.emplace_back([&, r]() {
            const void *rec = nullptr;
            uint64_t tot = 0;
            std::vector<uint64_t> cnt((size_t)world, 0);
            const int err = (scenario == 1 && r == victim) ? SPM_E_OVERFLOW : SPM_OK;
            const uint64_t mine = scenario == 3 ? (~0ull / record_bytes) - 5 : n[(size_t)r];
            status[(size_t)r] = gatherv_protocol(*T[(size_t)r], r, world, root, err, local[(size_t)r].data(), mine, record_bytes,
                                                &rec, &tot, cnt.data());
            if (r == root) {
                root_records = rec;
                root_total = tot;
                root_counts = cnt;
            }
        });
    for (std::thread &x : th)
        x.join();
    if (detail)
        for (int r = 0; r < world; ++r)
            detail[r] = status[(size_t)r];
    int verdict = F.timed_out ? SPM_E_HIP : SPM_OK;
    for (int r = 0; r < world && verdict == SPM_OK; ++r) {
        int want = SPM_OK;
        if (scenario == 1)
            want = r == victim ? SPM_E_OVERFLOW : SPM_E_PEER;
        else if (scenario == 2)
            want = r == root ? SPM_E_NOMEM : SPM_E_PEER;
        else if (scenario == 3)
            want = SPM_E_OVERFLOW;
        if (status[(size_t)r] != want)
            verdict = SPM_E_INVALID;
    }
    if (verdict == SPM_OK && (scenario == 0 || scenario == 4)) {
        uint64_t at = 0, total = 0;
        for (int r = 0; r < world; ++r)
            total += n[(size_t)r];
        if (root_total != total || (total && !root_records))
            verdict = SPM_E_INVALID;
        for (int r = 0; r < world && verdict == SPM_OK; ++r) {
            if (root_counts[(size_t)r] != n[(size_t)r] ||
                (n[(size_t)r] && memcmp((const uint8_t *)root_records + at, local[(size_t)r].data(), (size_t)(n[(size_t)r] * record_bytes)) != 0))
                verdict = SPM_E_INVALID;
            at += n[(size_t)r] * record_bytes;
        }
    }
    for (loopback_transport *t : T)
        delete t;
    return verdict;
}

} // namespace spm_hip
