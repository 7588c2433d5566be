// sharded_build.cpp — ONE global map from a cloud sharded over several GPUs, from a plain C++ host (no Python, no
// torch.distributed): one process per GPU, RCCL called inside libgndt (gndt_build_global_device, include/gndt.h).
//
//   sharded_build <cloud.f32> <points> <gridLen> <zLen> <slope_interval> <rank> <world> <id_file> [device]
//
// <cloud.f32> holds `points` packed xyz records; point 0 is the origin (receiver.cpp:145) and is not binned; rank r takes
// the contiguous range r of the remaining points.  Rank 0 writes the communicator's 128-byte id to <id_file> (the other
// ranks wait for it): any side channel does.  Every rank ends with the map of the whole cloud and prints its size.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "gndt.h"

#define CHECK(call)                                                                                  \
    do {                                                                                             \
        const int rc__ = (call);                                                                     \
        if (rc__ != GNDT_OK) { std::printf("ERROR %d at %s: %s\n", rc__, #call, h ? gndt_last_error(h) : gndt_comm_last_error()); return 1; } \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 9) { std::printf("usage: %s cloud.f32 points gridLen zLen slope_interval rank world id_file [device [global|owner]]\n", argv[0]); return 2; }
    const size_t points = std::strtoull(argv[2], nullptr, 10);
    const int rank = std::atoi(argv[6]), world = std::atoi(argv[7]);
    const int device = argc > 9 ? std::atoi(argv[9]) : rank;
    gndt_handle* h = nullptr;
    // the communicator's id: made by rank 0, read by the others
    char id[GNDT_COMM_ID_BYTES];
    if (rank == 0) {
        CHECK(gndt_comm_unique_id(id));
        std::string tmp = std::string(argv[8]) + ".tmp";
        FILE* f = std::fopen(tmp.c_str(), "wb");
        if (!f || std::fwrite(id, 1, sizeof id, f) != sizeof id) { std::printf("ERROR: cannot write %s\n", tmp.c_str()); return 1; }
        std::fclose(f);
        std::rename(tmp.c_str(), argv[8]);
    } else {
        FILE* f = nullptr;
        for (int tries = 0; tries < 600 && !(f = std::fopen(argv[8], "rb")); ++tries) std::this_thread::sleep_for(std::chrono::milliseconds(100));
        if (!f || std::fread(id, 1, sizeof id, f) != sizeof id) { std::printf("ERROR: cannot read %s\n", argv[8]); return 1; }
        std::fclose(f);
    }
    gndt_comm* comm = nullptr;
    CHECK(gndt_comm_create(id, rank, world, device, &comm));

    // this rank's contiguous range of the binned points (indices 1 .. points-1 of the file)
    const size_t nb = points - 1, lo = 1 + rank * nb / world, hi = 1 + (rank + 1) * nb / world, n = hi - lo;
    std::vector<float> shard(3 * (n ? n : 1));
    float origin[3];
    FILE* f = std::fopen(argv[1], "rb");
    if (!f || std::fread(origin, 4, 3, f) != 3 || std::fseek(f, (long)(lo * 12), SEEK_SET) != 0 ||
        std::fread(shard.data(), 12, n, f) != n) { std::printf("ERROR: cannot read %s\n", argv[1]); return 1; }
    std::fclose(f);

    gndt_params P;
    std::memset(&P, 0, sizeof P);
    P.grid_len = std::strtof(argv[3], nullptr); P.z_len = std::strtof(argv[4], nullptr); P.slope_interval = std::strtof(argv[5], nullptr);
    P.demand = GNDT_DEMAND_SLOPE; P.min_points = 3; P.device_id = device; P.strategy = GNDT_STRATEGY_AUTO;
    CHECK(gndt_create(&P, &h));
    CHECK(gndt_set_origin(h, origin));
    void* d_shard = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipMalloc(&d_shard, shard.size() * 4) != hipSuccess ||
        hipMemcpy(d_shard, shard.data(), shard.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { std::printf("ERROR: device copy\n"); return 1; }
    if (argc > 10 && std::string(argv[10]) == "owner") {
        // the points travel: every rank ends with the columns it owns and the global row of each of its rows
        gndt_owned_info info;
        const uint32_t* global_row = nullptr;           // device pointer: row of local row r in the map of the whole cloud
        CHECK(gndt_build_owned_device(h, comm, d_shard, n, 12, lo - 1, nb, &global_row, &info, nullptr));
        std::printf("rank %d/%d: shard %zu points -> owns %llu points, %llu nodes; global map: nodes %llu columns %llu slopes %llu  "
                    "(split %.3f ms, exchange %.3f ms [%llu B out], build %.3f ms, order %.3f ms)\n", rank, world, n,
                    (unsigned long long)info.owned_points, (unsigned long long)info.local_nodes, (unsigned long long)info.global_nodes,
                    (unsigned long long)info.global_columns, (unsigned long long)info.global_slopes, info.split_ms, info.exchange_ms,
                    (unsigned long long)info.bytes_sent, info.build_ms, info.order_ms);
        (void)hipFree(d_shard);
        gndt_destroy(h);
        gndt_comm_destroy(comm);
        return 0;
    }
    gndt_exchange_times t;
    CHECK(gndt_build_global_device(h, comm, d_shard, n, 12, lo - 1, nb, &t, nullptr));
    uint64_t nodes = 0, columns = 0, slopes = 0;
    CHECK(gndt_sync(h, &nodes, &columns, &slopes));
    std::printf("rank %d/%d: shard %zu points -> %llu local nodes; global map: nodes %llu columns %llu slopes %llu  "
                "(shard %.3f ms, exchange %.3f ms, finalize %.3f ms)\n", rank, world, n, (unsigned long long)t.local_nodes,
                (unsigned long long)nodes, (unsigned long long)columns, (unsigned long long)slopes, t.shard_ms, t.exchange_ms, t.finalize_ms);
    (void)hipFree(d_shard);
    gndt_destroy(h);
    gndt_comm_destroy(comm);
    return 0;
}
