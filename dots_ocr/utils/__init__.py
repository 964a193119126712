from dots_ocr_amd.prompts import dict_promptmode_to_prompt  # noqa: F401
