// rccl_shim.cpp -- TEST ONLY. A stand-in for librccl that lets several processes on ONE GPU run the collective part of
// ufomap_map_insert_batch (ufomap_hip.hip loads whatever UFOMAP_RCCL_LIB names and binds ncclGetUniqueId /
// ncclCommInitRank / ncclAllGather / ncclCommDestroy / ncclGetErrorString). The "network" is a POSIX shared-memory
// segment: an all-gather stages every rank's slot through it between two barriers. Stream semantics are kept the blunt
// way -- the stream is synchronised, the copies are synchronous -- which is all the tests need.
//   build: hipcc -O2 -shared -fPIC tests/cpp/rccl_shim.cpp -o tests/cpp/librccl_shim.so -lrt
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace
{
constexpr size_t kSlotMax = 48u << 20;  // bytes per rank in the segment
struct Header {
	std::atomic<uint32_t> arrived, generation, attached;
};
struct Comm {
	bool replicate;  // UFOMAP_SHIM_REPLICATE=1 (scripts/bench_batch_model.py): ONE process plays all ranks -- every slot of an
	                 // all-gather is a copy of the caller's (device-to-device on the stream): the walk for N ranks' scans, measured on one GPU
	int world, rank, fd;
	char name[64];
	uint8_t* base;
	size_t bytes;
	Header* hdr() { return reinterpret_cast<Header*>(base); }
	uint8_t* slot(int r) { return base + 4096 + (size_t)r * kSlotMax; }
};
void barrier(Comm* c)
{
	Header* h = c->hdr();
	const uint32_t gen = h->generation.load(std::memory_order_acquire);
	if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->world) {
		h->arrived.store(0, std::memory_order_relaxed);
		h->generation.store(gen + 1, std::memory_order_release);
	} else {
		while (h->generation.load(std::memory_order_acquire) == gen) usleep(50);
	}
}
}  // namespace

extern "C" {
struct ShimId {
	char b[128];
};
int ncclGetUniqueId(void* id)
{
	memset(id, 0, 128);
	snprintf(static_cast<char*>(id), 64, "/ufoshim_%d_%ld", (int)getpid(), (long)time(nullptr));
	return 0;
}
int ncclCommInitRank(void** comm, int world, ShimId id, int rank)
{
	Comm* c = new Comm;
	c->world = world;
	c->rank = rank;
	const char* rep = getenv("UFOMAP_SHIM_REPLICATE");
	c->replicate = rep && *rep && '0' != *rep;
	if (c->replicate) {
		c->fd = -1;
		c->base = nullptr;
		c->bytes = 0;
		*comm = c;
		return 0;
	}
	snprintf(c->name, sizeof(c->name), "%s", id.b);
	c->bytes = 4096 + (size_t)world * kSlotMax;
	c->fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
	if (c->fd < 0 || ftruncate(c->fd, (off_t)c->bytes) != 0) return 2;
	c->base = static_cast<uint8_t*>(mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, c->fd, 0));
	if (c->base == MAP_FAILED) return 2;
	// (a fresh segment is zero-filled: the counters start at 0) wait until every rank is attached
	c->hdr()->attached.fetch_add(1);
	while (c->hdr()->attached.load() < (uint32_t)world) usleep(100);
	*comm = c;
	return 0;
}
int ncclCommDestroy(void* comm)
{
	Comm* c = static_cast<Comm*>(comm);
	if (c->replicate) {
		delete c;
		return 0;
	}
	munmap(c->base, c->bytes);
	close(c->fd);
	shm_unlink(c->name);
	delete c;
	return 0;
}
int ncclAllGather(const void* send, void* recv, size_t count, int /*datatype: bytes*/, void* comm, hipStream_t stream)
{
	Comm* c = static_cast<Comm*>(comm);
	if (c->replicate) {
		for (int r = 0; r < c->world; ++r)
			if (hipMemcpyAsync(static_cast<uint8_t*>(recv) + (size_t)r * count, send, count, hipMemcpyDeviceToDevice, stream) != hipSuccess) return 1;
		return 0;
	}
	if (count > kSlotMax) return 3;
	if (hipStreamSynchronize(stream) != hipSuccess) return 1;
	if (hipMemcpy(c->slot(c->rank), send, count, hipMemcpyDeviceToHost) != hipSuccess) return 1;
	barrier(c);
	for (int r = 0; r < c->world; ++r)
		if (hipMemcpy(static_cast<uint8_t*>(recv) + (size_t)r * count, c->slot(r), count, hipMemcpyHostToDevice) != hipSuccess) return 1;
	barrier(c);  // (nobody overwrites its slot before everybody has read it)
	return 0;
}
const char* ncclGetErrorString(int e) { return 0 == e ? "ok" : (1 == e ? "shim: HIP error" : (2 == e ? "shim: shared memory" : "shim: slot too large")); }
}
