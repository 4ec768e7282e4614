// libspm/hip/communicator.hpp -- C++ face of the multi-GPU exchange step (include/spm_hip.h, spm_hip_comm_* /
// spm_hip_gatherv_hits): one process per GPU, every rank scans its text shard with
// spm_scan_opts{left_context = 1, pos_offset = shard begin} (the shard rule of SURVEY.md 8(e)), then the hit records are
// gathered to one rank over RCCL.  The reference has no counterpart: its matchers are single-process
// (/root/reference/libspm/libspm/matcher/seqan_pattern_base.hpp:40-52).
#pragma once

#include <array>
#include <cstring>
#include <vector>

#include <hip/hip_runtime_api.h>

#include <libspm/hip/context.hpp>

namespace spm::hip
{
using unique_id = std::array<char, 128>;

// rank 0 makes the id and hands it to the other ranks out of band (MPI_Bcast, a file, a socket ...)
inline unique_id make_unique_id() noexcept
{
    unique_id id{};
    if (spm_hip_comm_unique_id(id.data()) != SPM_OK)
        fatal("spm_hip_comm_unique_id", default_context());
    return id;
}

class communicator
{
    spm_comm * _comm{nullptr};
    int _rank{0}, _world{1};

public:
    communicator(unique_id const & id, int rank, int world) noexcept : _rank{rank}, _world{world}
    {
        if (spm_hip_comm_init(default_context(), id.data(), rank, world, &_comm) != SPM_OK)
            fatal("spm_hip_comm_init", default_context());
    }
    communicator(communicator const &) = delete;
    communicator & operator=(communicator const &) = delete;
    ~communicator() { spm_hip_comm_destroy(_comm); }

    int rank() const noexcept { return _rank; }
    int world() const noexcept { return _world; }

    // this rank's share [lo, hi) of a text of n symbols: contiguous, 1 KiB aligned (device loads stay aligned)
    std::pair<std::uint64_t, std::uint64_t> shard(std::uint64_t n) const noexcept
    {
        std::uint64_t per = (n + static_cast<std::uint64_t>(_world) - 1) / static_cast<std::uint64_t>(_world);
        per = (per + 1023) / 1024 * 1024;
        std::uint64_t const lo = std::min<std::uint64_t>(n, static_cast<std::uint64_t>(_rank) * per);
        return {lo, std::min<std::uint64_t>(n, lo + per)};
    }

    // every rank's hits on `root`, rank order = ascending shard order (per shard: arrival order); empty elsewhere.
    // Failure is collective (include/spm_hip.h): if any rank cannot contribute -- its scan overflowed its hit buffer, the
    // root cannot hold the records -- EVERY rank gets a status back (the failing rank its own, the others SPM_E_PEER)
    // and nothing was sent; no rank is left waiting in a collective.
    int try_gatherv(spm_hits * local, std::vector<spm_hit> & out, int root = 0) const noexcept
    {
        void const * d = nullptr;
        std::uint64_t n = 0;
        out.clear();
        int const rc = spm_hip_gatherv_hits(_comm, local, root, &d, &n, nullptr);
        if (rc != SPM_OK)
            return rc;
        out.resize(n);
        if (n && hipMemcpy(out.data(), d, n * sizeof(spm_hit), hipMemcpyDeviceToHost) != hipSuccess)
            return SPM_E_HIP;
        return SPM_OK;
    }

    // the same for callers that treat any failure as fatal (the noexcept style of the matcher call operators)
    std::vector<spm_hit> gatherv(spm_hits * local, int root = 0) const noexcept
    {
        std::vector<spm_hit> out;
        if (try_gatherv(local, out, root) != SPM_OK)
            fatal("spm_hip_gatherv_hits", default_context());
        return out;
    }
};
} // namespace spm::hip
