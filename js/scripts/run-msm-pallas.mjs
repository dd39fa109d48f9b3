// Mirror of scripts/run-msm-pallas.ts:  node js/scripts/run-msm-pallas.mjs <n> [gpus] [--evaluate] [--json]
import { pallasParams } from "../concrete/params.mjs";
import { main } from "./msm-drivers.mjs";
main(pallasParams, "unsafe").catch((e) => { console.error(e); process.exit(1); });
