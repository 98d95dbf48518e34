// host_affinity.cpp -- the host side of "one process per GPU": how many host threads a rank starts and which CPUs they
// run on (no GPU code; the device's PCI address comes from kbbq_hip.hip).
//
// Under torch.distributed.run every rank of a node runs the same host stages (scan / fill / format are 99 % of the file
// path's wall time): each gets usable CPUs / LOCAL_WORLD_SIZE threads (host_threads.h), and its threads are bound to the
// CPUs of the NUMA node its GPU hangs on, so that the page-locked staging slabs a rank fills are local to the PCIe root
// complex that uploads them.
#include "../../include/kbbq_hip.h"
#include "host_threads.h"
#include "raw_vector.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include <sched.h>

int kbbq_set_error_(int code, const char* msg);      // defined in kbbq_hip.hip

// "0-3,8,10-11" -> set; returns the number of CPUs named
static int parse_cpulist(const char* s, cpu_set_t* set)
{
    CPU_ZERO(set);
    int n = 0;
    while (*s) {
        while (*s == ',' || *s == ' ' || *s == '\n') ++s;
        if (!*s) break;
        char* e = nullptr;
        const long a = strtol(s, &e, 10);
        if (e == s) break;
        long b = a;
        s = e;
        if (*s == '-') { b = strtol(s + 1, &e, 10); if (e == s + 1) break; s = e; }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) if (c >= 0 && !CPU_ISSET((int)c, set)) { CPU_SET((int)c, set); ++n; }
    }
    return n;
}

static bool read_text(const std::string& path, char* buf, size_t cap)
{
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return false;
    const size_t k = fread(buf, 1, cap - 1, f);
    fclose(f);
    buf[k] = 0;
    return k > 0;
}

extern "C" {

int kbbq_host_threads(size_t work_bytes) { return (int)kbbq_threads_for(work_bytes); }

int kbbq_host_advise_huge(void* p, size_t bytes) { if (p && bytes) kbbq_advise_huge(p, bytes); return KBBQ_OK; }

// Bind the calling thread -- and so every thread it starts from now on: the library's readers, packers and writers are
// started per call -- to the CPUs of the NUMA node of PCI device `pci_bus_id` ("0000:c1:00.0", as hipDeviceGetPCIBusId
// prints it), intersected with the CPUs the process may use already.  *numa_node: the node, or -1 when the system names
// none (single-socket hosts, most virtual machines) -- nothing is changed then, nor when the intersection is empty.
// *ncpus: CPUs in the mask afterwards.  The sysfs root is /sys (KBBQ_SYSFS_ROOT: the tests' stand-in tree).
int kbbq_bind_host_to_pci(const char* pci_bus_id, int* numa_node, int* ncpus)
{
    if (!pci_bus_id) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_bind_host_to_pci: NULL bus id");
    (void)kbbq_usable_cpus();                       // cache the job-wide figure before the mask narrows (host_threads.h)
    cpu_set_t now;
    CPU_ZERO(&now);
    if (sched_getaffinity(0, sizeof now, &now) != 0) return kbbq_set_error_(KBBQ_E_ARG, "sched_getaffinity failed");
    if (numa_node) *numa_node = -1;
    if (ncpus) *ncpus = CPU_COUNT(&now);
    const char* root = getenv("KBBQ_SYSFS_ROOT");
    const std::string sys = root && *root ? root : "/sys";
    std::string id(pci_bus_id);
    for (auto& ch : id) if (ch >= 'A' && ch <= 'F') ch = (char)(ch - 'A' + 'a');        // sysfs names are lower case
    char buf[4096];
    if (!read_text(sys + "/bus/pci/devices/" + id + "/numa_node", buf, sizeof buf)) return KBBQ_OK;
    const int node = atoi(buf);
    if (node < 0) return KBBQ_OK;
    if (!read_text(sys + "/devices/system/node/node" + std::to_string(node) + "/cpulist", buf, sizeof buf)) return KBBQ_OK;
    cpu_set_t of_node, both;
    if (parse_cpulist(buf, &of_node) == 0) return KBBQ_OK;
    CPU_AND(&both, &of_node, &now);
    if (CPU_COUNT(&both) == 0) return KBBQ_OK;
    if (sched_setaffinity(0, sizeof both, &both) != 0) return KBBQ_OK;                   // not permitted: stay as we are
    if (numa_node) *numa_node = node;
    if (ncpus) *ncpus = CPU_COUNT(&both);
    return KBBQ_OK;
}

} // extern "C"
