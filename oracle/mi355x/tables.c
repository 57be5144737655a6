/* TEST INFRASTRUCTURE ONLY -- see tables.h: the extended name tables and the factory switches that know the GPU ids. */
#define OCR_MI355X_TABLES_IMPL
#include "tables.h"

#define MI355X_MAX_NAMES 64
const char *guid_types_mi355x[MI355X_MAX_NAMES], *schedulerHeuristic_types_mi355x[MI355X_MAX_NAMES], *workpile_types_mi355x[MI355X_MAX_NAMES],
           *commplatform_types_mi355x[MI355X_MAX_NAMES];

static void extend(const char **to, const char **stock, int n_stock, const char *name) {
    int i;
    ASSERT(n_stock + 2 <= MI355X_MAX_NAMES);
    for(i = 0; i < n_stock; ++i) to[i] = stock[i];
    to[n_stock] = name;
    to[n_stock + 1] = NULL;
}
/* before main: the machine description is parsed long after */
__attribute__((constructor)) static void mi355xTables(void) {
    extend(guid_types_mi355x, guid_types, guidMax_id, "GPUHASH");
    extend(schedulerHeuristic_types_mi355x, schedulerHeuristic_types, schedulerHeuristicMax_id, "GPU");
    extend(workpile_types_mi355x, workpile_types, workpileMax_id, "GPU");
    extend(commplatform_types_mi355x, commplatform_types, commPlatformMax_id, "RCCL");
}

ocrGuidProviderFactory_t *newGuidProviderFactoryMi355x(guidType_t type, ocrParamList_t *typeArg) {
    if((int)type == guidGpuHash_id) return newGuidProviderFactoryGpuHash(typeArg, (u32)type);
    return newGuidProviderFactory(type, typeArg);
}
ocrSchedulerHeuristicFactory_t *newSchedulerHeuristicFactoryMi355x(schedulerHeuristicType_t type, ocrParamList_t *perType) {
    if((int)type == schedulerHeuristicGpu_id) return newOcrSchedulerHeuristicFactoryGpu(perType, (u32)type);
    return newSchedulerHeuristicFactory(type, perType);
}
ocrWorkpileFactory_t *newWorkpileFactoryMi355x(workpileType_t type, ocrParamList_t *perType) {
    if((int)type == workpileGpu_id) return newOcrWorkpileFactoryGpu(perType);
    return newWorkpileFactory(type, perType);
}
ocrCommPlatformFactory_t *newCommPlatformFactoryMi355x(commPlatformType_t type, ocrParamList_t *typeArg) {
    if((int)type == commPlatformRccl_id) return newCommPlatformFactoryRccl(typeArg);
    return newCommPlatformFactory(type, typeArg);
}
