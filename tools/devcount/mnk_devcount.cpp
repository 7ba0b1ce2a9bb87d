// Device-wide hardware counters around a region of GPU work WITHOUT dispatch serialization: rocprofiler-sdk's device
// counting service (agent-scoped sampling).  `rocprofv3 --pmc` serializes kernel dispatches, which the task-DAG schedule's
// two persistent kernels (pivot chain + bulk kernel, waiting for each other) cannot survive; this tool library is loaded
// into the measured process (ROCP_TOOL_LIBRARIES=<this .so>) and driven from it through three C functions:
//     mnk_devcount_start("COUNTER_A,COUNTER_B")   -> 0 / error;  counting starts
//     mnk_devcount_sample(names, values, cap)     -> number of counters; values = sums over all dimension instances since start
//     mnk_devcount_stop()
// One counter set per start (the hardware has 8 SQ / 4 TCC / 2 GRBM slots per pass: MI355X_MICROARCH.md).
// Build: tools/devcount/build.sh.  Driver: tools/devcount_dag.py.
#include <rocprofiler-sdk/registration.h>
#include <rocprofiler-sdk/rocprofiler.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace {
rocprofiler_context_id_t g_ctx = {};
rocprofiler_buffer_id_t g_buf = {};
rocprofiler_agent_id_t g_agent = {};
rocprofiler_counter_config_id_t g_profile = {.handle = 0};
bool g_ready = false, g_running = false;
std::map<std::string, rocprofiler_counter_id_t> g_counters;   // name -> id (supported on the agent)
std::map<uint64_t, std::string> g_names;                      // id -> name
std::string g_err;
size_t g_expected = 0;

#define RP(call)                                                                                          \
    do {                                                                                                  \
        rocprofiler_status_t st_ = (call);                                                                \
        if (st_ != ROCPROFILER_STATUS_SUCCESS) {                                                          \
            g_err = std::string(#call) + ": " + rocprofiler_get_status_string(st_);                       \
            fprintf(stderr, "mnk_devcount: %s\n", g_err.c_str());                                         \
            return -1;                                                                                    \
        }                                                                                                 \
    } while (0)

int tool_init(rocprofiler_client_finalize_t, void*) {
    // first GPU agent
    std::vector<rocprofiler_agent_v0_t> agents;
    rocprofiler_query_available_agents_cb_t cb = [](rocprofiler_agent_version_t, const void** arr, size_t n, void* ud) {
        auto* v = static_cast<std::vector<rocprofiler_agent_v0_t>*>(ud);
        for (size_t i = 0; i < n; ++i) {
            const auto* a = static_cast<const rocprofiler_agent_v0_t*>(arr[i]);
            if (a->type == ROCPROFILER_AGENT_TYPE_GPU) v->push_back(*a);
        }
        return ROCPROFILER_STATUS_SUCCESS;
    };
    RP(rocprofiler_query_available_agents(ROCPROFILER_AGENT_INFO_VERSION_0, cb, sizeof(rocprofiler_agent_t), &agents));
    if (agents.empty()) { g_err = "no GPU agent"; return -1; }
    g_agent = agents[0].id;
    RP(rocprofiler_create_context(&g_ctx));
    RP(rocprofiler_create_buffer(g_ctx, 4096, 2048, ROCPROFILER_BUFFER_POLICY_LOSSLESS,
                                 [](rocprofiler_context_id_t, rocprofiler_buffer_id_t, rocprofiler_record_header_t**, size_t, void*, uint64_t) {},
                                 nullptr, &g_buf));
    auto th = rocprofiler_callback_thread_t{};
    RP(rocprofiler_create_callback_thread(&th));
    RP(rocprofiler_assign_callback_thread(g_buf, th));
    RP(rocprofiler_configure_device_counting_service(
        g_ctx, g_buf, g_agent,
        [](rocprofiler_context_id_t ctx, rocprofiler_agent_id_t, rocprofiler_device_counting_agent_cb_t set_config, void*) {
            if (g_profile.handle != 0) set_config(ctx, g_profile);
        },
        nullptr));
    g_ready = true;
    return 0;
}

void tool_fini(void*) {}

int load_counters() {
    if (!g_counters.empty()) return 0;
    std::vector<rocprofiler_counter_id_t> ids;
    RP(rocprofiler_iterate_agent_supported_counters(
        g_agent,
        [](rocprofiler_agent_id_t, rocprofiler_counter_id_t* c, size_t n, void* ud) {
            auto* v = static_cast<std::vector<rocprofiler_counter_id_t>*>(ud);
            for (size_t i = 0; i < n; ++i) v->push_back(c[i]);
            return ROCPROFILER_STATUS_SUCCESS;
        },
        &ids));
    for (auto& id : ids) {
        rocprofiler_counter_info_v0_t info;
        RP(rocprofiler_query_counter_info(id, ROCPROFILER_COUNTER_INFO_VERSION_0, &info));
        g_counters.emplace(info.name, id);
        g_names.emplace(id.handle, info.name);
    }
    return 0;
}
}  // namespace

extern "C" {

int mnk_devcount_available(void) { return g_ready ? 1 : 0; }
const char* mnk_devcount_error(void) { return g_err.c_str(); }

int mnk_devcount_start(const char* names_csv) {
    if (!g_ready) { g_err = "tool not initialized (ROCP_TOOL_LIBRARIES must name this library before the HIP runtime loads)"; return -1; }
    if (g_running) { g_err = "already counting"; return -1; }
    if (load_counters()) return -1;
    std::vector<rocprofiler_counter_id_t> want;
    g_expected = 0;
    std::string all(names_csv ? names_csv : "");
    size_t pos = 0;
    while (pos < all.size()) {
        size_t end = all.find(',', pos);
        if (end == std::string::npos) end = all.size();
        const std::string name = all.substr(pos, end - pos);
        pos = end + 1;
        auto it = g_counters.find(name);
        if (it == g_counters.end()) { g_err = "counter not supported on this agent: " + name; fprintf(stderr, "mnk_devcount: %s\n", g_err.c_str()); return -2; }
        want.push_back(it->second);
        rocprofiler_counter_info_v1_t info;
        RP(rocprofiler_query_counter_info(it->second, ROCPROFILER_COUNTER_INFO_VERSION_1, &info));
        g_expected += info.dimensions_instances_count;
    }
    if (want.empty()) { g_err = "no counters"; return -1; }
    RP(rocprofiler_create_counter_config(g_agent, want.data(), want.size(), &g_profile));
    RP(rocprofiler_start_context(g_ctx));
    g_running = true;
    return 0;
}

// names_out: cap * 64 bytes (NUL-terminated names, 64 bytes apart); values: cap doubles.  Returns the number of counters.
int mnk_devcount_sample(char* names_out, double* values, int cap) {
    if (!g_running) { g_err = "not counting"; return -1; }
    std::vector<rocprofiler_counter_record_t> rec(g_expected + 64);
    size_t n = rec.size();
    RP(rocprofiler_sample_device_counting_service(g_ctx, {}, ROCPROFILER_COUNTER_FLAG_NONE, rec.data(), &n));
    std::map<std::string, double> sums;
    for (size_t i = 0; i < n; ++i) {
        rocprofiler_counter_id_t cid = {.handle = 0};
        rocprofiler_query_record_counter_id(rec[i].id, &cid);
        auto it = g_names.find(cid.handle);
        sums[it == g_names.end() ? std::string("?") : it->second] += rec[i].counter_value;
    }
    int k = 0;
    for (auto& [name, v] : sums) {
        if (k >= cap) break;
        snprintf(names_out + 64 * k, 64, "%s", name.c_str());
        values[k] = v;
        ++k;
    }
    return k;
}

int mnk_devcount_stop(void) {
    if (!g_running) return 0;
    g_running = false;
    RP(rocprofiler_stop_context(g_ctx));
    return 0;
}

rocprofiler_tool_configure_result_t* rocprofiler_configure(uint32_t, const char*, uint32_t, rocprofiler_client_id_t* id) {
    id->name = "mnk_devcount";
    static auto cfg = rocprofiler_tool_configure_result_t{sizeof(rocprofiler_tool_configure_result_t), &tool_init, &tool_fini, nullptr};
    return &cfg;
}
}
