// Mirror of scripts/run-msm-pallas-projective.ts:  node js/scripts/run-msm-pallas-projective.mjs <n> [gpus] [--evaluate] [--json]
import { pallasParams } from "../concrete/params.mjs";
import { main } from "./msm-drivers.mjs";
main(pallasParams, "projective").catch((e) => { console.error(e); process.exit(1); });
