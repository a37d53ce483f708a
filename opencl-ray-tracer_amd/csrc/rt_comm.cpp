// Multi-GPU exchange step of the C ABI: one process drives n GPUs (one context each); after every GPU has summed its
// own seed shards into its accumulation buffer, ONE all-reduce(sum) of the width*height*3 float buffers over RCCL
// (xGMI) gives every GPU the frame's sum.  The reference has a single device and no exchange step
// (src/cpp/main.cpp:27-29,175); this is the only collective of the path.
//
// RCCL is loaded on first use (dlopen), so the library itself loads on machines without it and a process that already
// holds an RCCL (PyTorch's) keeps using that one.
#include "../../include/ocl_rt_amd.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <mutex>
#include <string>
#include <vector>

void wbrt_set_error_(const char *msg);

namespace {

int fail(int code, const std::string &msg)
{
	wbrt_set_error_(msg.c_str());
	return code;
}

// the subset of the NCCL/RCCL C API used here (rccl.h: ncclComm_t is an opaque pointer, ncclResult_t an int enum with
// ncclSuccess = 0, ncclFloat32 = 7, ncclSum = 0)
typedef void *comm_t;
typedef int result_t;
struct Rccl
{
	void *handle = nullptr;
	result_t (*CommInitAll)(comm_t *, int, const int *) = nullptr;
	result_t (*CommDestroy)(comm_t) = nullptr;
	result_t (*AllReduce)(const void *, void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
	result_t (*Reduce)(const void *, void *, size_t, int, int, int, comm_t, hipStream_t) = nullptr;
	result_t (*GroupStart)() = nullptr;
	result_t (*GroupEnd)() = nullptr;
	const char *(*GetErrorString)(result_t) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mutex;

// Loads the library from the first of `names` that opens and resolves the entry points into r.
bool load_rccl_from(const char *const *names, int n_names, Rccl &r, std::string &why)
{
	void *h = nullptr;
	std::string last;
	for (int i = 0; i < n_names && !h; ++i)
	{
		h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
		if (!h)
		{
			const char *e = dlerror(); // one call: it clears the message it returns
			last = e ? e : "dlopen failed";
		}
	}
	if (!h) { why = "RCCL not found: " + last; return false; }
	r.CommInitAll = (decltype(r.CommInitAll))dlsym(h, "ncclCommInitAll");
	r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
	r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
	r.Reduce = (decltype(r.Reduce))dlsym(h, "ncclReduce");
	r.GroupStart = (decltype(r.GroupStart))dlsym(h, "ncclGroupStart");
	r.GroupEnd = (decltype(r.GroupEnd))dlsym(h, "ncclGroupEnd");
	r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
	if (!r.CommInitAll || !r.CommDestroy || !r.AllReduce || !r.Reduce || !r.GroupStart || !r.GroupEnd)
	{
		why = "RCCL library lacks a required symbol";
		dlclose(h);
		return false;
	}
	r.handle = h;
	return true;
}

bool load_rccl(std::string &why)
{
	std::lock_guard<std::mutex> lock(g_rccl_mutex);
	if (g_rccl.handle) return true;
	const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
	Rccl r;
	if (!load_rccl_from(names, 3, r, why)) return false;
	g_rccl = r;
	return true;
}

std::string rccl_error(const char *what, result_t rc)
{
	std::string s = std::string(what) + " failed (" + std::to_string(rc) + ")";
	if (g_rccl.GetErrorString) s += std::string(": ") + g_rccl.GetErrorString(rc);
	return s;
}

} // namespace

struct wbrt_comm
{
	std::vector<int> devices;
	std::vector<comm_t> comms;
};

extern "C" {

int wbrt_comm_create(const int32_t *devices, int32_t n, wbrt_comm **out)
{
	if (!out) return fail(WBRT_ERR_INVALID, "null out pointer");
	*out = nullptr;
	if (!devices || n <= 0) return fail(WBRT_ERR_INVALID, "no devices");
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(WBRT_ERR_NO_DEVICE, "no HIP device available");
	for (int i = 0; i < n; ++i)
	{
		if (devices[i] < 0 || devices[i] >= count) return fail(WBRT_ERR_INVALID, "device ordinal out of range");
		for (int j = 0; j < i; ++j)
			if (devices[j] == devices[i]) return fail(WBRT_ERR_INVALID, "a device appears twice (RCCL wants one rank per GPU)");
	}
	wbrt_comm *c = new wbrt_comm();
	c->devices.assign(devices, devices + n);
	if (n > 1) // a single GPU needs no communicator: the all-reduce is the identity
	{
		std::string why;
		if (!load_rccl(why)) { delete c; return fail(WBRT_ERR_NO_DEVICE, why); }
		c->comms.assign((size_t)n, nullptr);
		const result_t rc = g_rccl.CommInitAll(c->comms.data(), n, c->devices.data());
		if (rc != 0) { delete c; return fail(WBRT_ERR_HIP, rccl_error("ncclCommInitAll", rc)); }
	}
	*out = c;
	return WBRT_OK;
}

int wbrt_comm_destroy(wbrt_comm *comm)
{
	if (!comm) return WBRT_OK;
	for (size_t i = 0; i < comm->comms.size(); ++i)
		if (comm->comms[i])
		{
			(void)hipSetDevice(comm->devices[i]);
			(void)g_rccl.CommDestroy(comm->comms[i]);
		}
	delete comm;
	return WBRT_OK;
}

int32_t wbrt_comm_size(const wbrt_comm *comm) { return comm ? (int32_t)comm->devices.size() : 0; }

// Test hook (not in the public header): what wbrt_comm_create does when RCCL cannot be loaded, with the library name
// given by the caller (a name that does not exist forces the failure).  Never replaces a library already in use.
int wbrt_rccl_probe_(const char *library_name)
{
	if (!library_name) return fail(WBRT_ERR_INVALID, "null library name");
	Rccl r;
	std::string why;
	if (!load_rccl_from(&library_name, 1, r, why)) return fail(WBRT_ERR_NO_DEVICE, why);
	return WBRT_OK; // the handle stays open: a loaded RCCL is never unloaded
}

int wbrt_comm_reduce_sum(wbrt_comm *comm, void *const *device_buffers, size_t count, int32_t root, void *const *streams)
{
	if (!comm || !device_buffers) return fail(WBRT_ERR_INVALID, "null argument");
	const size_t n = comm->devices.size();
	if (root < 0 || (size_t)root >= n) return fail(WBRT_ERR_INVALID, "root out of range");
	for (size_t i = 0; i < n; ++i)
		if (!device_buffers[i]) return fail(WBRT_ERR_INVALID, "null device buffer");
	if (n == 1) return WBRT_OK;
	result_t rc = g_rccl.GroupStart();
	if (rc != 0) return fail(WBRT_ERR_HIP, rccl_error("ncclGroupStart", rc));
	for (size_t i = 0; i < n; ++i)
	{
		rc = g_rccl.Reduce(device_buffers[i], device_buffers[i], count, /*ncclFloat32*/ 7, /*ncclSum*/ 0, root, comm->comms[i],
		                   streams ? (hipStream_t)streams[i] : nullptr);
		if (rc != 0) { (void)g_rccl.GroupEnd(); return fail(WBRT_ERR_HIP, rccl_error("ncclReduce", rc)); }
	}
	rc = g_rccl.GroupEnd();
	if (rc != 0) return fail(WBRT_ERR_HIP, rccl_error("ncclGroupEnd", rc));
	return WBRT_OK;
}

int wbrt_comm_allreduce_sum(wbrt_comm *comm, void *const *device_buffers, size_t count, void *const *streams)
{
	if (!comm || !device_buffers) return fail(WBRT_ERR_INVALID, "null argument");
	const size_t n = comm->devices.size();
	for (size_t i = 0; i < n; ++i)
		if (!device_buffers[i]) return fail(WBRT_ERR_INVALID, "null device buffer");
	if (n == 1) return WBRT_OK;
	result_t rc = g_rccl.GroupStart();
	if (rc != 0) return fail(WBRT_ERR_HIP, rccl_error("ncclGroupStart", rc));
	for (size_t i = 0; i < n; ++i)
	{
		rc = g_rccl.AllReduce(device_buffers[i], device_buffers[i], count, /*ncclFloat32*/ 7, /*ncclSum*/ 0, comm->comms[i],
		                      streams ? (hipStream_t)streams[i] : nullptr);
		if (rc != 0) { (void)g_rccl.GroupEnd(); return fail(WBRT_ERR_HIP, rccl_error("ncclAllReduce", rc)); }
	}
	rc = g_rccl.GroupEnd();
	if (rc != 0) return fail(WBRT_ERR_HIP, rccl_error("ncclGroupEnd", rc));
	return WBRT_OK;
}

} // extern "C"
